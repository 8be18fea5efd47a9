#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest22.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r03_gputest22.log
timeout -k 10 200 python tests/tools/soak.py 60 > gpurun_out/r03_soak22.txt 2>&1; echo "soak rc=$?"; tail -6 gpurun_out/r03_soak22.txt
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r03_bench22.json 2> gpurun_out/r03_bench22.err; echo "bench rc=$?"; cut -c1-700 gpurun_out/r03_bench22.json
