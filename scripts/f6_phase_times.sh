#!/bin/bash
# device time of phase 1 (loss-only launch) and phase 2 (resume launch) of fused6 for each variant library (tree = in-tree)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = tree ]; then L=tf_seq2seq_losses_amd/libctc_amd.so; else L=scratch/libctc_v_$v.so; fi
  OUT=gpurun_out/pt_$v; rm -rf $OUT; mkdir -p $OUT
  CTC_AMD_LIB=$L rocprofv3 --kernel-trace -d $OUT --output-format csv -- python3 scripts/phase_times.py > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "fused6" in r["Kernel_Name"]]
p1, p2 = d[10::2], d[11::2]   # skip the first 5 pairs
import statistics as st
print("$v: phase 1 %.1f us (min %.1f)  phase 2 %.1f us (min %.1f)  sum %.1f" % (st.mean(p1) / 1e3, min(p1) / 1e3, st.mean(p2) / 1e3, min(p2) / 1e3, (st.mean(p1) + st.mean(p2)) / 1e3))
PY
done
