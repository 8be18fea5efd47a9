#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_hvp.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -4
python scripts/hvp_time.py > gpurun_out/r03_hvp_time4.txt 2>&1; cat gpurun_out/r03_hvp_time4.txt
