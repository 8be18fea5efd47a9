#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_hvp.py tests/test_gpu_graph.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r03_gputest17.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r03_gputest17.log
[ $rc -eq 0 ] && timeout -k 10 300 python tests/tools/soak_hvp.py 40 > gpurun_out/r03_soak_hvp17.txt 2>&1; echo "soak rc=$?"; tail -5 gpurun_out/r03_soak_hvp17.txt
[ $rc -eq 0 ] && timeout -k 10 200 python scripts/hvp_time.py > gpurun_out/r03_hvp_time17.txt 2>&1; tail -12 gpurun_out/r03_hvp_time17.txt
