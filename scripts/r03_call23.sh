#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest23.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03_gputest23.log
AB_ROUNDS=10 timeout -k 10 300 python scripts/ab_time.py tree lazy > gpurun_out/r03_ab23_b256.txt 2>&1; tail -3 gpurun_out/r03_ab23_b256.txt
F6_B=64 AB_ROUNDS=10 timeout -k 10 300 python scripts/ab_time.py tree lazy > gpurun_out/r03_ab23_b64.txt 2>&1; tail -3 gpurun_out/r03_ab23_b64.txt
timeout -k 10 200 python tests/tools/soak.py 40 > gpurun_out/r03_soak23.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r03_soak23.txt
