# SQ counters of the loss kernel, phase 1 (loss-only call) and phase 2 (resume call) separately: two rocprofv3 --pmc passes of scripts/lossonly_once.py
# usage (on the GPU box): bash scripts/pmc_phase.sh   -> gpurun_out/pmc_phase/
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_phase
mkdir -p $OUT
rm -rf $OUT/p1 $OUT/p2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/p1 --output-format csv -- python3 scripts/lossonly_once.py > $OUT/p1.log 2>&1; echo "pass1 rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -d $OUT/p2 --output-format csv -- python3 scripts/lossonly_once.py > $OUT/p2.log 2>&1; echo "pass2 rc=$?"
python3 - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/pmc_phase/{d}/*/*_counter_collection.csv")
    if not f: print(d, "no csv"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "fused6" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)
    for i in ids[-4:]:
        print(d, "dispatch", i, {k: int(v) for k, v in per[i].items()})
PY
