#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest6.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r03_gputest6.log
python scripts/hvp_time.py > gpurun_out/r03_hvp_time3.txt 2>&1; cat gpurun_out/r03_hvp_time3.txt
