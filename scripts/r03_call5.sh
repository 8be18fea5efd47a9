#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python scripts/hvp_time.py > gpurun_out/r03_hvp_time2.txt 2>&1; cat gpurun_out/r03_hvp_time2.txt
timeout -k 10 300 python -m pytest tests/test_gpu_hvp.py tests/test_gpu_round3.py -m gpu -q 2>&1 | tail -3
for cfg in "--B 32 --V 4096" "--B 64 --V 2048" "--U 512" "--B 256 --T 255 --U 126 --V 32"; do
  tag=$(echo $cfg | tr -d ' -')
  rm -rf gpurun_out/r03_prof_$tag
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03_prof_$tag --output-format csv -- python3 bench.py $cfg --steps 50 --warmup 10 --no-secondary --no-cpu-baseline > gpurun_out/r03_prof_$tag.json 2> gpurun_out/r03_prof_$tag.err
  echo "== $cfg"
  python3 - "$tag" <<'PY'
import csv, glob, collections, sys
f = glob.glob(f"gpurun_out/r03_prof_{sys.argv[1]}/*/*_kernel_trace.csv")
if f:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]:
        v2 = sorted(v)
        print(f"{len(v):5d} x  median {v2[len(v2)//2]:9.1f} us  {k}")
PY
done
