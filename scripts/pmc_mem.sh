#!/bin/bash
# Dev: memory-path PMC passes (TLB, L1 stalls, L2 read latency, TA / LDS FIFOs) on the 64-image forward; each pass in its
# own run, counters only.  Usage: bash scripts/pmc_mem.sh  (from the repo root; writes gpurun_out/pmc_mem_summary.txt)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for pass in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_THRASHING_STALL_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    # (a fifth pass with the TA_* counters — TA_BUSY_avr, TA_ADDR_STALLED_BY_TC_CYCLES_sum, ... — did not finish on this pool: left out)
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmcm_$i -o pmc -- python3 $ROOT/scripts/prof_forward.py 64 2 > $OUT/pmcm_$i.log 2>&1 || echo "pass $i failed"
    echo "pass $i done" | tee -a $OUT/pmc_mem_progress.log
done
cd $ROOT
python3 scripts/pmc_summary.py $(find $OUT/pmcm_* -name '*counter_collection.csv') > $OUT/pmc_mem_summary.txt
rm -rf $OUT/pmcm_*
echo done
