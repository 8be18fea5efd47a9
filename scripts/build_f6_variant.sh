#!/bin/bash
# builds scratch/libctc_f6v.so: the library with ctc_fused6.hip (classic, NL=2) recompiled with extra flags, e.g.
#   scripts/build_f6_variant.sh -DCTC_F6_DEBUG      then CTC_AMD_LIB=scratch/libctc_f6v.so python scripts/...
# Variant builds define CTC_DIAG (the experiment switches of the sources exist only then); ctc_capi.hip is rebuilt with it
# too, because the diagnostic workspace regions are part of the layout it computes.
set -e
cd "$(dirname "$0")/.."
C=tf_seq2seq_losses_amd/csrc
mkdir -p scratch/objv
KIND=${F6_KIND:-0}; NLV=${F6_NL:-2}
NAME=$([ "$KIND" = 0 ] && echo classic || echo simplified)
TAG=$(basename ${F6_OUT:-f6v})
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -I$C -DCTC_FUSED_KIND=$KIND -DCTC_FUSED6_NL=$NLV -fno-slp-vectorize -DCTC_DIAG "$@" -c ${F6_SRC:-$C/ctc_fused6.hip} -o scratch/objv/$TAG.o &
[ -f scratch/objv/capi_diag.o ] && [ scratch/objv/capi_diag.o -nt $C/ctc_capi.hip ] && [ scratch/objv/capi_diag.o -nt $C/ctc_common.h ] || hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -I$C -DCTC_DIAG -c $C/ctc_capi.hip -o scratch/objv/capi_diag.o
wait
OBJS=$(ls $C/_obj/*.o | grep -v "ctc_fused6_${NAME}_nl${NLV}.o" | grep -v "ctc_capi.o" | grep -v "ctc_wide.o")
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS scratch/objv/capi_diag.o scratch/objv/$TAG.o -o ${F6_OUT:-scratch/libctc_f6v.so}
echo built ${F6_OUT:-scratch/libctc_f6v.so}
