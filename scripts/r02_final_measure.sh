#!/bin/bash
# round 2: everything that is quoted in DESIGN.md / profiles/ from ONE box: rocprof stats + PMC of the north-star kernel,
# the default bench line, the shape table, the README table, the length sweep, the autograd timing
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash scripts/r02_profile_head.sh r02_final > $OUT/profile.log 2>&1
echo "profile done"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done"
python3 scripts/autograd_time.py > $OUT/autograd_time.txt 2>&1
python3 scripts/ab_time.py tree > $OUT/ab_time.txt 2>/dev/null
python3 benchmarks/reference_table.py --json $OUT/ref_table.json > $OUT/ref_table.md 2> $OUT/ref_table.err
echo "ref table done"
python3 benchmarks/reference_table.py --tsweep --steps 5 --warmup 2 --json $OUT/tsweep.json > $OUT/tsweep.md 2> $OUT/tsweep.err
echo "sweep done"
python3 scripts/shape_table.py > $OUT/shape_table.md 2> $OUT/shape_table.err
echo "shape table done"
