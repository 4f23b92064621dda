#!/bin/bash
# Dev A/B of a lab library against the product one on the attention kernel: parity tests, the ramp leg, the headline.
# Usage: bash scripts/attn_ab.sh scripts/_lab/libX.so
set -e -o pipefail
LAB=$1
OUT=gpurun_out
mkdir -p $OUT
POPE_LIB_PATH=$LAB timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $OUT/attn_ab_test.log 2>&1 || { tail -20 $OUT/attn_ab_test.log; exit 1; }
tail -1 $OUT/attn_ab_test.log
for lib in "" $LAB "" $LAB; do
    POPE_LIB_PATH=$lib timeout -k 10 200 python - <<'PY' 2>/dev/null | sed "s|^|[${lib:-product}] |"
import json, subprocess, sys, torch
sys.path.insert(0, ".")
import bench_legs
r = bench_legs.attention_ramp_leg(torch.device("cuda:0"))
print(" ".join(f"{k}: {v['ms']:.3f} ms ({v['exact_pass_rate']:.2f})" for k, v in r["ramps_log2_units_per_tile"].items()))
PY
    POPE_LIB_PATH=$lib timeout -k 10 200 python bench.py --no-legs --no-config5 --no-strict-f32 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('   headline', b['value'], b['roofline']['avg_ms_per_launch'], b['verified'])" | sed "s|^|[${lib:-product}] |"
done
