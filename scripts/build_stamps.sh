#!/bin/bash
# Diagnostic build of libctc with -DCTC_FUSED_STAMPS (per-wavefront work / barrier-wait cycle counters) -> scratch/libctc_stamps.so
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -Itf_seq2seq_losses_amd/csrc"
C=tf_seq2seq_losses_amd/csrc
mkdir -p scratch
hipcc $F -fno-honor-nans -DCTC_FUSED_KIND=0 -DCTC_FUSED_STAMPS -c $C/ctc_fused5.hip -o scratch/f4c_stamp.o &
hipcc $F -fno-honor-nans -DCTC_FUSED_KIND=1 -DCTC_FUSED_STAMPS -c $C/ctc_fused5.hip -o scratch/f4s_stamp.o &
wait
hipcc --offload-arch=gfx950 -fPIC -shared $C/_obj/ctc_kernels.o $C/_obj/ctc_fused_classic.o $C/_obj/ctc_fused_simplified.o $C/_obj/ctc_fused4_classic.o $C/_obj/ctc_fused4_simplified.o scratch/f4c_stamp.o scratch/f4s_stamp.o $C/_obj/ctc_hessian.o $C/_obj/ctc_hvp.o $C/_obj/ctc_capi.o -o scratch/libctc_stamps.so
