#!/bin/bash
# first GPU call of round 3: memory probe, cache experiment, bench at the driver's settings and at the builder's, GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
scratch/r03_memprobe > gpurun_out/r03_memprobe.json 2> gpurun_out/r03_memprobe.err; echo "memprobe rc=$?"
python scripts/r03_phase_cache.py > gpurun_out/r03_phase_cache.json 2> gpurun_out/r03_phase_cache.err; echo "phase_cache rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_20_5.json 2> gpurun_out/r03_bench_20_5.err; echo "bench 20/5 rc=$?"
python bench.py --steps 20 --warmup 5 --prewarm-ms 0 --no-secondary --no-cpu-baseline > gpurun_out/r03_bench_20_5_noprewarm.json 2>> gpurun_out/r03_bench_20_5.err; echo "bench 20/5 noprewarm rc=$?"
python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/r03_bench_200_20.json 2> gpurun_out/r03_bench_200_20.err; echo "bench 200/20 rc=$?"
python scripts/r03_host_profile.py > gpurun_out/r03_host_profile.txt 2>&1; echo "host profile rc=$?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest1.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r03_gputest1.log
