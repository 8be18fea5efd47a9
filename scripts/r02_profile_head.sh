#!/bin/bash
# round 2: re-profile the north-star kernel at HEAD (stats + PMC passes), outputs under gpurun_out/r02_head/
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_head}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python3 scripts/lossgrad_once.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python3 scripts/lossgrad_once.py > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -d $OUT/pmc_sq --output-format csv -- python3 scripts/lossgrad_once.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_ic --output-format csv -- python3 scripts/lossgrad_once.py > $OUT/pmc_ic.log 2>&1 || echo "icache pass failed" >> $OUT/pmc_ic.log
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
