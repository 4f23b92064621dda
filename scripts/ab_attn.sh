#!/bin/bash
# Dev: same-box A/B of attention build variants: scripts/ab_attn.sh "<flagsA>" "<flagsB>"
set -e
cd "$(dirname "$0")/../pope_amd/csrc"
for flags in "$1" "$2" "$1" "$2"; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DATTN_STAMPS $flags -c attention_f16x3.hip -o attention_f16x3.o 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o libpope_hip.so gemm_f32.o gemm_f16x3.o layernorm.o attention_f32.o attention_f16x3.o match.o capi.o
    echo "== flags: [$flags]"
    python3 ../../scripts/attn_stamps.py | tail -2
done
