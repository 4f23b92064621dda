#!/bin/bash
# Dev: lab copies of the library that differ only in the LN-fused GEMM's build flags (tile geometry -DRL_GEO=2|3|96, staging
# issue -DRL_SPREAD, phase stamps -DRL_STAMPS): scripts/rowln_ab.sh NAME "<flags>" [NAME "<flags>" ...]
# -> scripts/_lab/libpope_NAME.so (git-ignored; travels to the GPU box).  Time them with scripts/ab_bench.sh / rowln_stamps.py.
set -e
cd "$(dirname "$0")/../pope_amd/csrc"
make -j8 >/dev/null 2>&1
mkdir -p ../../scripts/_lab
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    T=$(mktemp -d)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c gemm_rowln.hip -o $T/r.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/_lab/libpope_$name.so $(ls *.o | grep -v "^gemm_rowln.o") $T/r.o
    rm -rf $T
    echo "built scripts/_lab/libpope_$name.so [$flags]"
done
