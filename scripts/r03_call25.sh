#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_ROUNDS=10 timeout -k 10 300 python scripts/ab_time.py ctl x1y3 x0y4 x1y4 seg > gpurun_out/r03_ab25_b256.txt 2>&1; tail -5 gpurun_out/r03_ab25_b256.txt
F6_B=64 AB_ROUNDS=10 timeout -k 10 300 python scripts/ab_time.py ctl x1y3 x0y4 x1y4 seg > gpurun_out/r03_ab25_b64.txt 2>&1; tail -5 gpurun_out/r03_ab25_b64.txt
