#!/bin/bash
# Dev: lab copies of the library that differ only in attention_f16.hip's build flags (-DAF_LAB=<bits>: timing-only ablations):
# scripts/attn_f16_ab.sh NAME "<flags>" ... -> scripts/_lab/libpope_NAME.so; time them with scripts/attn_f16_time.py
set -e
cd "$(dirname "$0")/../pope_amd/csrc"
make -j8 >/dev/null 2>&1
mkdir -p ../../scripts/_lab
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    T=$(mktemp -d)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c attention_f16.hip -o $T/a.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/_lab/libpope_$name.so $(ls *.o | grep -v "^attention_f16.o") $T/a.o
    rm -rf $T
    echo "built scripts/_lab/libpope_$name.so [$flags]"
done
