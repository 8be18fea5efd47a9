#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest10.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_gputest10.log
# two ranks sharing the one GPU, gloo collectives: rehearsal of the N > 1 loop of bench.py (pipeline depth 2, four sum buffers)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/r03_bench_gloo2.json 2> gpurun_out/r03_bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 1200 gpurun_out/r03_bench_gloo2.json; tail -3 gpurun_out/r03_bench_gloo2.err
python bench.py --steps 20 --warmup 5 --emulate-collective --no-secondary --no-cpu-baseline > gpurun_out/r03_bench_emulated.json 2> gpurun_out/r03_bench_emulated.err; echo "emulated rc=$?"; tail -c 900 gpurun_out/r03_bench_emulated.json
