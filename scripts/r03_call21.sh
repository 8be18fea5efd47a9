#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_ROUNDS=12 timeout -k 10 400 python scripts/ab_time.py ctl x2y3 x3y2 x3y3 rn6 rn6x2y3 base > gpurun_out/r03_ab21_b256.txt 2>&1; tail -8 gpurun_out/r03_ab21_b256.txt
F6_B=64 AB_ROUNDS=12 timeout -k 10 400 python scripts/ab_time.py ctl x2y3 x3y2 x3y3 rn6 rn6x2y3 base > gpurun_out/r03_ab21_b64.txt 2>&1; tail -8 gpurun_out/r03_ab21_b64.txt
