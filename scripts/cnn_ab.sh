#!/bin/bash
# Dev A/B of a lab library against the product one on the LoFTR CNN: parity tests, then the backbone's time, alternating.
# Usage: bash scripts/cnn_ab.sh scripts/_lab/libX.so
set -e -o pipefail
LAB=$1
OUT=gpurun_out
mkdir -p $OUT
POPE_LIB_PATH=$LAB timeout -k 10 300 python -m pytest tests/test_gpu_loftr.py -x -q > $OUT/cnn_ab_test.log 2>&1 || { tail -20 $OUT/cnn_ab_test.log; exit 1; }
tail -1 $OUT/cnn_ab_test.log
for lib in "" $LAB "" $LAB; do
    POPE_LIB_PATH=$lib timeout -k 10 120 python scripts/loftr_time.py 2>/dev/null | grep "backbone" | sed "s|^|[${lib:-product}] |"
done
