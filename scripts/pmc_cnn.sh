#!/bin/bash
# HBM traffic of ONE ResNet-FPN call (the LoFTR leg's roofline kernel sequence): two rocprofv3 --pmc passes (counters only,
# no trace domains beyond the kernel trace) around scripts/cnn_trace.py, summed over the dispatches of its last call.
# Usage: bash scripts/pmc_cnn.sh   (from the repo root on the GPU box; writes gpurun_out/pmc_cnn.json)
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pairs in 3 24; do
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_cnn_${pairs}_$c -o pmc -- python3 $ROOT/scripts/cnn_trace.py $pairs > $OUT/pmc_cnn_${pairs}_$c.log 2>&1
        echo "pmc pass $pairs pairs $c done"
    done
done
cd $ROOT
python3 scripts/pmc_cnn_sum.py $OUT > $OUT/pmc_cnn.json
rm -rf $OUT/pmc_cnn_3_* $OUT/pmc_cnn_24_*
cat $OUT/pmc_cnn.json
