#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest4.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r03_gputest4.log
python scripts/hvp_time.py > gpurun_out/r03_hvp_time.txt 2>&1; cat gpurun_out/r03_hvp_time.txt
rm -rf gpurun_out/r03_hvptrace && rocprofv3 --kernel-trace --stats -d gpurun_out/r03_hvptrace --output-format csv -- python3 scripts/hvp_time.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r03_hvptrace/*/*_kernel_trace.csv")
if f:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v2 = sorted(v)
        print(f"{len(v):5d} x  median {v2[len(v2)//2]:9.1f} us  min {v2[0]:9.1f}  {k}")
PY
