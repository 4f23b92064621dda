#!/bin/bash
# PMC passes on the SAM ViT-H encoder (each counter group in its own run, counters + kernel trace only — never combined
# with other trace domains).  Usage: bash scripts/sam_pmc.sh [f16x3|f16]   (from the repo root; writes gpurun_out/sam_pmc_*.txt)
set -e -o pipefail
PREC=${1:-f16x3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    timeout -k 10 420 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/sampmc_$tag -o pmc -- python3 $ROOT/scripts/sam_time.py vit_h 2 1 $PREC > $OUT/sampmc_$tag.log 2>&1
    echo "pmc pass $tag done"
done
cd $ROOT
python3 scripts/pmc_summary.py $(find $OUT/sampmc_* -name '*counter_collection.csv') > $OUT/sam_pmc_$PREC.txt
rm -rf $OUT/sampmc_*
echo "sam_pmc done"
