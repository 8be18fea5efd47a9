#!/bin/bash
# quick kernel-time check: rocprofv3 kernel stats of bench.py (north-star config), prints the fused kernels' averages
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-qprof}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ${@:2} > $OUT/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "fused" in r["Name"] or "order" in r["Name"] or "reduce" in r["Name"]:
        print(r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
tail -1 $OUT/stats.log | cut -c1-400
