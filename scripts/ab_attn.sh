#!/bin/bash
# Dev: same-box A/B of attention build variants: scripts/ab_attn.sh "<flagsA>" "<flagsB>" ...   (each timed twice;
# STAMPS=1 adds -DATTN_STAMPS and prints the per-phase cycle breakdown instead)
set -e
cd "$(dirname "$0")/../pope_amd/csrc"
# the variant builds go to a scratch library (POPE_LIB_PATH), never over the product libpope_hip.so / its objects
OUT=$(mktemp -d)
trap 'rm -rf "$OUT"' EXIT
for rep in 1 2; do
for flags in "$@"; do
    key=$(echo "$flags" | tr -c 'A-Za-z0-9' '_')
    if [ ! -f $OUT/$key.so ]; then
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off ${STAMPS:+-DATTN_STAMPS} $flags -c attention_f16x3.hip -o $OUT/$key.o 2>/dev/null
        hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$key.so $(ls *.o | grep -v "^attention_f16x3.o") $OUT/$key.o
    fi
    echo "== flags: [$flags]"
    if [ -n "$STAMPS" ]; then POPE_LIB_PATH=$OUT/$key.so python3 ../../scripts/attn_stamps.py | tail -2
    else POPE_LIB_PATH=$OUT/$key.so python3 ../../scripts/attn_time.py | tail -1; fi
done
done
