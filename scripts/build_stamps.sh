#!/bin/bash
# Diagnostic build of libctc with -DCTC_FUSED_STAMPS (per-wavefront work / barrier-wait cycle counters) -> scratch/libctc_stamps.so
exec "$(dirname "$0")/build_variant.sh" stamps "-DCTC_FUSED_STAMPS"
