#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcmem
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum -d $OUT/p1 --output-format csv -- python3 scripts/lossonly_once.py > $OUT/p1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum -d $OUT/p2 --output-format csv -- python3 scripts/lossonly_once.py > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_CYCLE_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -d $OUT/p3 --output-format csv -- python3 scripts/lossonly_once.py > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if "fused6" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in acc.items():
        ids = sorted(v)
        print(d, k, "per dispatch (alternating phase 1 / phase 2):", [round(v[i]) for i in ids][:6])
PY
tail -2 $OUT/p1.log
