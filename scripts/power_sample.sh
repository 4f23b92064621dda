#!/bin/bash
# Dev: sample rocm-smi power / clocks while the bench step runs (is the 1.7-1.9 GHz of finding 6 the power cap?).
# Usage: bash scripts/power_sample.sh   (GPU box; prints a few samples)
set -e
mkdir -p gpurun_out
python bench.py --steps 300 --warmup 2 --no-legs --no-config5 --no-strict-f32 --no-cpu-baseline --no-verify > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 16
for i in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction\|edge" | tr '\n' ';'
    echo
    sleep 0.7
done
wait $BP
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
python -c "import json;b=json.load(open('gpurun_out/power_bench.json'));print('bench', b['value'])"
