#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest2.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r03_gputest2.log
scratch/r03_memprobe > gpurun_out/r03_memprobe2.json 2> gpurun_out/r03_memprobe2.err; echo "memprobe rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench2_20_5.json 2> gpurun_out/r03_bench2_20_5.err; echo "bench 20/5 rc=$?"
python scripts/r03_host_profile.py > gpurun_out/r03_host_profile2.txt 2>&1; echo "host profile rc=$?"
timeout -k 10 200 python tests/tools/soak.py 90 > gpurun_out/r03_soak1.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r03_soak1.log
rm -rf gpurun_out/r03_trace && rocprofv3 --kernel-trace --stats -d gpurun_out/r03_trace --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/r03_bench2_prof.json 2> gpurun_out/r03_bench2_prof.err; echo "rocprof rc=$?"
python3 - <<'PY'
import csv, glob, statistics as st
f = glob.glob("gpurun_out/r03_trace/*/*_kernel_trace.csv")
if f:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0])) if "fused6" in r["Kernel_Name"]]
    d2 = sorted(d[50:])
    print("fused6 launches", len(d), "after the first 50: min %.1f median %.1f mean %.1f p90 %.1f p99 %.1f max %.1f us" % (d2[0], st.median(d2), st.mean(d2), d2[int(len(d2)*.9)], d2[int(len(d2)*.99)], d2[-1]))
    print("first 30:", [round(x, 1) for x in d[:30]])
PY
