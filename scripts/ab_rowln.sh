#!/bin/bash
# Dev: stamped builds of the LN-fused GEMM (scripts/rowln_stamps.py): scripts/ab_rowln.sh "<flagsA>" "<flagsB>" ...
set -e
cd "$(dirname "$0")/../pope_amd/csrc"
OUT=$(mktemp -d)
trap 'rm -rf "$OUT"' EXIT
for flags in "$@"; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DRL_STAMPS $flags -c gemm_rowln.hip -o $OUT/r.o 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib.so $(ls *.o | grep -v "^gemm_rowln.o") $OUT/r.o
    echo "== flags: [$flags]"
    POPE_LIB_PATH=$OUT/lib.so python3 ../../scripts/rowln_stamps.py 2>/dev/null | grep "tile 1"
done
