#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_ROUNDS=12 timeout -k 10 400 python scripts/ab_time.py tree p0 p0b1 b1 x3y2 x2y3 pfd3 base > gpurun_out/r03_ab19_b256.txt 2>&1; tail -9 gpurun_out/r03_ab19_b256.txt
F6_B=64 AB_ROUNDS=12 timeout -k 10 400 python scripts/ab_time.py tree p0 p0b1 b1 x3y2 x2y3 pfd3 base > gpurun_out/r03_ab19_b64.txt 2>&1; tail -9 gpurun_out/r03_ab19_b64.txt
