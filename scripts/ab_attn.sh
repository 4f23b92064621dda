#!/bin/bash
# Dev: same-box A/B of attention build variants: scripts/ab_attn.sh "<flagsA>" "<flagsB>"
set -e
cd "$(dirname "$0")/../pope_amd/csrc"
# the stamped builds go to a scratch library (POPE_LIB_PATH), never over the product libpope_hip.so / its objects
OUT=$(mktemp -d)
trap 'rm -rf "$OUT"' EXIT
for flags in "$1" "$2" "$1" "$2"; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DATTN_STAMPS $flags -c attention_f16x3.hip -o $OUT/attention_f16x3.o 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libpope_hip.so gemm_f32.o gemm_f16x3.o layernorm.o attention_f32.o $OUT/attention_f16x3.o match.o capi.o
    echo "== flags: [$flags]"
    POPE_LIB_PATH=$OUT/libpope_hip.so python3 ../../scripts/attn_stamps.py | tail -2
done
