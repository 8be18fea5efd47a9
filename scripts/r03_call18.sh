#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_sweep.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -q -x > gpurun_out/r03_gputest18.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r03_gputest18.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python scripts/ab_time.py tree base > gpurun_out/r03_ab18_b256.txt 2>&1; tail -4 gpurun_out/r03_ab18_b256.txt
F6_B=64 timeout -k 10 200 python scripts/ab_time.py tree base > gpurun_out/r03_ab18_b64.txt 2>&1; tail -4 gpurun_out/r03_ab18_b64.txt
