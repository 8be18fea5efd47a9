#!/bin/bash
# runs scripts/f6_check.py <mode> against each variant library given as argument (names under scratch/libctc_v_<name>.so)
mode=$1; shift
for v in "$@"; do
  echo "=== variant $v"
  F6_STAMPS=1 CTC_AMD_LIB=scratch/libctc_v_$v.so timeout -k 10 120 python scripts/f6_check.py $mode 2>&1 | grep -v amdgpu.ids | cut -c1-220
done
