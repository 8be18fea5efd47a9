#!/bin/bash
# scripts/build_variant.sh NAME "-DFLAG ..." : diagnostic build of libctc with extra flags for ctc_fused5.hip -> scratch/libctc_NAME.so
# (the six fused5 units -- kind x label positions per lane -- are rebuilt with the flags; everything else comes from csrc/_obj)
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -Itf_seq2seq_losses_amd/csrc"
C=tf_seq2seq_losses_amd/csrc
mkdir -p scratch
OBJS=""
for k in 0 1; do for nl in 1 2 4; do
  hipcc $F -fno-honor-nans -DCTC_FUSED_KIND=$k -DCTC_FUSED5_NL=$nl $2 -c $C/ctc_fused5.hip -o scratch/f5_${k}_${nl}_$1.o &
  OBJS="$OBJS scratch/f5_${k}_${nl}_$1.o"
done; done
wait
hipcc --offload-arch=gfx950 -fPIC -shared $C/_obj/ctc_kernels.o $C/_obj/ctc_fused_classic.o $C/_obj/ctc_fused_simplified.o $C/_obj/ctc_fused4_classic.o $C/_obj/ctc_fused4_simplified.o $OBJS $C/_obj/ctc_hessian.o $C/_obj/ctc_hvp.o $C/_obj/ctc_capi.o -o scratch/libctc_$1.so
