#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_ROUNDS=10 timeout -k 10 300 python scripts/ab_time.py tree la1 seg rn6 > gpurun_out/r03_ab24_b256.txt 2>&1; tail -4 gpurun_out/r03_ab24_b256.txt
F6_B=64 AB_ROUNDS=10 timeout -k 10 300 python scripts/ab_time.py tree la1 seg rn6 > gpurun_out/r03_ab24_b64.txt 2>&1; tail -4 gpurun_out/r03_ab24_b64.txt
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest24.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r03_gputest24.log
