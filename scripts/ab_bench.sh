#!/bin/bash
# Dev: same-box A/B of whole library builds inside bench.py (the f16x3 step alone, self-check on): for every library
# given, ROUNDS runs of the headline step; one summary line per run (pairs/s, ms per step, verified, per-kernel ms).
# Usage: bash scripts/ab_bench.sh [-r ROUNDS] scripts/_lab/libpope_A.so scripts/_lab/libpope_B.so ...
set -e -o pipefail
ROUNDS=2
if [ "$1" = "-r" ]; then ROUNDS=$2; shift 2; fi
mkdir -p gpurun_out
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    POPE_LIB_PATH=$(realpath $lib) timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-f32 --no-legs --no-config5 \
        > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { grep -q '"metric"' gpurun_out/ab_tmp.json || { echo "$lib FAILED"; tail -5 gpurun_out/ab_tmp.err; exit 1; }; }   # (a lab build that fails the self-check still prints its line)
    python3 - "$lib" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
k = {e['kernel'].replace('gemm_', ''): e['avg_ms'] for e in d.get('kernels', [])}
print(f"{sys.argv[1].split('/')[-1]:24s} {d['value']:8.1f} pairs/s {d['ms_per_step']:7.2f} ms verified={d.get('verified')} " +
      ' '.join(f"{n}={v:.4f}" for n, v in k.items()), flush=True)
PY
  done
done
