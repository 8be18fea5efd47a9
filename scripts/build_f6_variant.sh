#!/bin/bash
# builds scratch/libctc_f6v.so: the library with ctc_fused6.hip (classic, NL=2) recompiled with extra flags, e.g.
#   scripts/build_f6_variant.sh -DCTC_F6_DEBUG      then CTC_AMD_LIB=scratch/libctc_f6v.so python scripts/...
set -e
cd "$(dirname "$0")/.."
C=tf_seq2seq_losses_amd/csrc
mkdir -p scratch/objv
KIND=${F6_KIND:-0}; NLV=${F6_NL:-2}
NAME=$([ "$KIND" = 0 ] && echo classic || echo simplified)
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -I$C -DCTC_FUSED_KIND=$KIND -DCTC_FUSED6_NL=$NLV -fno-slp-vectorize "$@" -c ${F6_SRC:-$C/ctc_fused6.hip} -o scratch/objv/$(basename ${F6_OUT:-f6v}).o
OBJS=$(ls $C/_obj/*.o | grep -v "ctc_fused6_${NAME}_nl${NLV}.o")
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS scratch/objv/$(basename ${F6_OUT:-f6v}).o -o ${F6_OUT:-scratch/libctc_f6v.so}
echo built ${F6_OUT:-scratch/libctc_f6v.so}
