#!/bin/bash
# kernel-level profile of the Hessian-vector product and of the three-kernel pipeline at the north-star shape
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/hvp_prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 scripts/hvp_time.py > $OUT/log.txt 2>&1
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("  %-60s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
tail -3 $OUT/log.txt
