#!/bin/bash
# round 4: everything that is quoted in DESIGN.md / profiles/ from ONE box.  usage: bash scripts/r04_final_measure.sh [part]
# parts: 1 = bench lines, rocprof stats, PMC traffic, memory probes, HVP;  2 = tables (reference table, length sweep, shapes), soak
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_final
mkdir -p $OUT
PART=${1:-1}
COMMIT=$(cat .commit_id 2>/dev/null || echo unknown)
if [ "$PART" = 1 ]; then
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_settings.json 2> $OUT/bench_driver_settings.err; echo "bench 20/5 rc=$?"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
rm -rf $OUT/trace && rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "rocprof rc=$?"
python3 - <<'PY'
import csv, glob, statistics as st
f = glob.glob("gpurun_out/r04_final/trace/*/*_kernel_trace.csv")
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0])) if "fused6" in r["Kernel_Name"]]
t = sorted(d[-300:-100])  # (order of launches: warm-up steps, pre-warm, the 200 timed steps, 100 more with an event pair around each)
open("gpurun_out/r04_final/fused6_durations.txt", "w").write(
    "rocprofv3 --kernel-trace of `python3 bench.py --steps 200 --warmup 20`: fused6_kernel<0, 2, 4, 12, 1, 0>\n"
    f"all {len(d)} launches (incl. the pre-warm): first 40 (us): {[round(x) for x in d[:40]]}\n"
    f"the 200 timed launches: min {t[0]:.1f} median {st.median(t):.1f} mean {st.mean(t):.1f} p90 {t[int(len(t) * .9)]:.1f} max {t[-1]:.1f} us\n")
print(open("gpurun_out/r04_final/fused6_durations.txt").read())
PY
cp $OUT/trace/*/*_kernel_stats.csv $OUT/fused6_kernel_stats.csv
rm -rf $OUT/pmc_f $OUT/pmc_w
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f --output-format csv -- python3 scripts/lossgrad_once.py > $OUT/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w --output-format csv -- python3 scripts/lossgrad_once.py > $OUT/pmc_w.log 2>&1
python3 scripts/pmc_traffic.py $OUT/pmc_f/*/*_counter_collection.csv $OUT/pmc_w/*/*_counter_collection.csv fused6 $OUT/fused6_pmc_traffic.json $COMMIT > /dev/null; echo "pmc rc=$?"
rm -rf $OUT/pmc_hf $OUT/pmc_hw
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_hf --output-format csv -- python3 scripts/hvp_once.py > $OUT/pmc_hf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_hw --output-format csv -- python3 scripts/hvp_once.py > $OUT/pmc_hw.log 2>&1
python3 scripts/pmc_traffic.py $OUT/pmc_hf/*/*_counter_collection.csv $OUT/pmc_hw/*/*_counter_collection.csv hvp_fused $OUT/hvp_fused_pmc_traffic.json $COMMIT > /dev/null; echo "pmc hvp rc=$?"
python3 scripts/r03_phase_cache.py > $OUT/phase_cache.json 2> $OUT/phase_cache.err; echo "phase cache rc=$?"
python3 scripts/hvp_time.py > $OUT/hvp_time.txt 2>&1; echo "hvp time rc=$?"
python3 scripts/sharp_check.py 1 2 3 4 > $OUT/sharp_check.txt 2>&1; echo "sharp check rc=$?"
python3 scripts/phase_ab.py tree > $OUT/phase_times.txt 2>&1; echo "phase times rc=$?"
else
python3 benchmarks/reference_table.py --steps 100 --warmup 20 --json $OUT/ref_table.json > $OUT/ref_table.md 2> $OUT/ref_table.err; echo "ref table rc=$?"
python3 benchmarks/reference_table.py --tsweep --steps 5 --warmup 2 --json $OUT/tsweep.json > $OUT/tsweep.md 2> $OUT/tsweep.err; echo "sweep rc=$?"
python3 scripts/shape_table.py > $OUT/shape_table.md 2> $OUT/shape_table.err; echo "shape table rc=$?"
python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank gloo rc=$?"
fi
