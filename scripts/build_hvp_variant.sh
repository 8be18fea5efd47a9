#!/bin/bash
# builds scratch/libctc_hvpv.so: the product library with ctc_hvp_fused.hip (classic) recompiled with extra flags, e.g.
#   scripts/build_hvp_variant.sh -DCTC_HVPF_DIAG_NOLOAD    then CTC_AMD_LIB=scratch/libctc_hvpv.so python scripts/hvp_time.py
set -e
cd "$(dirname "$0")/.."
C=tf_seq2seq_losses_amd/csrc
mkdir -p scratch/objv
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -I$C -DCTC_FUSED_KIND=0 -fno-slp-vectorize "$@" -c $C/ctc_hvp_fused.hip -o scratch/objv/hvpv.o
OBJS=$(ls $C/_obj/*.o | grep -v "ctc_hvp_fused_classic.o")
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS scratch/objv/hvpv.o -o ${HVP_OUT:-scratch/libctc_hvpv.so}
echo built ${HVP_OUT:-scratch/libctc_hvpv.so}
