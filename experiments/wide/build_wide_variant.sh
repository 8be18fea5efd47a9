#!/bin/bash
# builds scratch/libctc_wide_diag.so: the product library plus the EXPERIMENTAL one-launch wide-vocabulary tier (ctc_wide.hip, DESIGN.md
# 5.2b; not a product unit), with ctc_capi.hip recompiled with -DCTC_DIAG -DCTC_WIDE_EXPERIMENT so that ctc_amd_debug_override("pipeline", "wide") selects it
# and ctc_amd_debug_override("wide", "diagN") sets its timing diagnostics.  Then e.g.
#   CTC_AMD_LIB=scratch/libctc_wide_diag.so python experiments/wide/wide_time.py 32,1000,128,4096 diag4 diag8 diag16 diag24
#   CTC_AMD_LIB=scratch/libctc_wide_diag.so python -m pytest experiments/wide/wide_checks.py -q
#   CTC_AMD_LIB=scratch/libctc_wide_diag.so python tests/tools/soak.py 300 wide
# (CTC_DIAG changes the workspace layout of ctc_common.h for the fused tiers' own diagnostics: use this library for the wide tier only.)
set -e
cd "$(dirname "$0")/../.."
C=tf_seq2seq_losses_amd/csrc
mkdir -p scratch/objv
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -I$C -DCTC_DIAG -DCTC_WIDE_EXPERIMENT"
hipcc $F -c experiments/wide/ctc_wide.hip -o scratch/objv/wide_diag.o &
hipcc $F -c $C/ctc_capi.hip -o scratch/objv/capi_diag.o &
wait
OBJS=$(ls $C/_obj/*.o | grep -v "ctc_wide.o" | grep -v "ctc_capi.o")  # (ctc_wide.o: a leftover of earlier builds)
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS scratch/objv/capi_diag.o scratch/objv/wide_diag.o -o scratch/libctc_wide_diag.so
echo built scratch/libctc_wide_diag.so
