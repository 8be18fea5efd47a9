#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest3.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r03_gputest3.log
python scripts/ab_time.py base tree > gpurun_out/r03_ab_trim.txt 2>&1; cat gpurun_out/r03_ab_trim.txt | tail -3
F6_B=64 python scripts/ab_time.py base tree > gpurun_out/r03_ab_trim_b64.txt 2>&1; cat gpurun_out/r03_ab_trim_b64.txt | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench3_20_5.json 2> gpurun_out/r03_bench3_20_5.err; echo "bench 20/5 rc=$?"
