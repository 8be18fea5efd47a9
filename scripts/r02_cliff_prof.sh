#!/bin/bash
# kernel-level profile of the shapes that miss the fused tier (wide vocabularies, long labels)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "--V 2048" "--V 8192 --B 32" "--U 512" "--U 300"; do
  name=$(echo $cfg | tr -d ' -')
  OUT=gpurun_out/cliff_$name; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $cfg > $OUT/log.txt 2>&1
  f=$(ls $OUT/*/*kernel_stats.csv | head -1)
  echo "== $cfg"; cut -d, -f1-4 $f | cut -c1-150 | head -8
done
