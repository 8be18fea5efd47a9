#!/bin/bash
# scripts/build_hess_variant.sh NAME "-DFLAG ..." : diagnostic build of libctc with extra flags for ctc_hessian.hip -> scratch/libctc_NAME.so
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -Itf_seq2seq_losses_amd/csrc"
C=tf_seq2seq_losses_amd/csrc
mkdir -p scratch
hipcc $F $2 -c $C/ctc_hessian.hip -o scratch/hess_$1.o
hipcc --offload-arch=gfx950 -fPIC -shared $C/_obj/ctc_kernels.o $C/_obj/ctc_fused_classic.o $C/_obj/ctc_fused_simplified.o $C/_obj/ctc_fused4_classic.o $C/_obj/ctc_fused4_simplified.o $C/_obj/ctc_fused5_classic_nl1.o $C/_obj/ctc_fused5_classic_nl2.o $C/_obj/ctc_fused5_classic_nl4.o $C/_obj/ctc_fused5_simplified_nl1.o $C/_obj/ctc_fused5_simplified_nl2.o $C/_obj/ctc_fused5_simplified_nl4.o scratch/hess_$1.o $C/_obj/ctc_hvp.o $C/_obj/ctc_capi.o -o scratch/libctc_$1.so
