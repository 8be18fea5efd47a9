#!/bin/bash
# wall time per call of the north-star shape for each variant library (and the in-tree library as "tree")
for v in tree "$@"; do
  if [ $v = tree ]; then L=tf_seq2seq_losses_amd/libctc_amd.so; else L=scratch/libctc_v_$v.so; fi
  echo -n "$v: "
  CTC_AMD_LIB=$L timeout -k 10 120 python scripts/f6_check.py nsc 2>&1 | grep fused | sed 's/.*grad/grad/' | cut -c1-120
done
