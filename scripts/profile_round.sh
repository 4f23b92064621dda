#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the bench command, then PMC passes (each in its own
# run, counters only — never combined with trace domains) on the 64-image forward driver.
# Usage: bash scripts/profile_round.sh   (from the repo root; writes under gpurun_out/)
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (a) the default command as the driver runs it (self-check launches and the strict-fp32 leg included), (b) the f16x3 step
# alone: per-kernel averages that can be compared with the line's HIP-event figures
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_default -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_profiled_default.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-verify --no-legs --no-config5 > $OUT/bench_profiled.json
# the secondary legs on their own (bench_legs.py): the LoFTR Matcher at 3 and 24 pairs, and one query of the drivers' loop
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_loftr -o loftr -- python3 $ROOT/bench.py --only loftr_matcher > $OUT/bench_loftr_matcher.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_driver -o driver -- python3 $ROOT/bench.py --only driver_step > $OUT/bench_driver_step.json
echo "stats passes done"
if [ -n "$STATS_ONLY" ]; then
    mkdir -p $OUT/profile_round
    cp $(find $OUT/prof_stats -name '*kernel_stats.csv') $OUT/profile_round/bench_kernel_stats.csv
    cp $(find $OUT/prof_stats_default -name '*kernel_stats.csv') $OUT/profile_round/bench_default_kernel_stats.csv
    cp $(find $OUT/prof_stats_loftr -name '*kernel_stats.csv') $OUT/profile_round/loftr_kernel_stats.csv
    cp $(find $OUT/prof_stats_driver -name '*kernel_stats.csv') $OUT/profile_round/driver_step_kernel_stats.csv
    mv $OUT/bench_profiled.json $OUT/bench_profiled_default.json $OUT/bench_loftr_matcher.json $OUT/bench_driver_step.json $OUT/profile_round/
    rm -rf $OUT/prof_stats $OUT/prof_stats_default $OUT/prof_stats_loftr $OUT/prof_stats_driver
    echo "profile_round (stats only) done"; exit 0
fi
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
            "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    timeout -k 10 420 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 $ROOT/scripts/prof_forward.py 64 2 > $OUT/pmc_$tag.log
    echo "pmc pass $tag done"
done
cd $ROOT
python3 scripts/pmc_summary.py $(find $OUT/pmc_* -name '*counter_collection.csv') > $OUT/pmc_summary.txt
# keep the summaries only: the raw traces exceed what gpurun copies back
mkdir -p $OUT/profile_round
cp $(find $OUT/prof_stats -name '*kernel_stats.csv') $OUT/profile_round/bench_kernel_stats.csv
cp $(find $OUT/prof_stats_default -name '*kernel_stats.csv') $OUT/profile_round/bench_default_kernel_stats.csv
cp $(find $OUT/prof_stats_loftr -name '*kernel_stats.csv') $OUT/profile_round/loftr_kernel_stats.csv
cp $(find $OUT/prof_stats_driver -name '*kernel_stats.csv') $OUT/profile_round/driver_step_kernel_stats.csv
mv $OUT/bench_loftr_matcher.json $OUT/bench_driver_step.json $OUT/profile_round/
mv $OUT/pmc_summary.txt $OUT/bench_profiled.json $OUT/bench_profiled_default.json $OUT/profile_round/
rm -rf $OUT/prof_stats $OUT/prof_stats_default $OUT/prof_stats_loftr $OUT/prof_stats_driver $OUT/pmc_*
echo "profile_round done"
