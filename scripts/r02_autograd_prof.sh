#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/agprof
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 scripts/autograd_time.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]
tot = 0
for r in csv.DictReader(open(f)):
    tot += float(r["TotalDurationNs"])
    if float(r["Percentage"]) > 0.5:
        print(r["Name"][:70], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2))
print("total GPU busy per step (us):", tot / 110 / 1e3)
PY
grep "autograd.grad" $OUT/log.txt
