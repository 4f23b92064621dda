#!/bin/bash
# Dev: lab copies of the library with the X-stationary GEMM of scripts/gemm_xstat_lab.hip dispatched for QKV / FC1 (same-box A/B
# against the product's 128 x 128 tile kernel): scripts/xstat_ab.sh NAME "<flags>" [NAME "<flags>" ...]
# -> scripts/_lab/libpope_NAME.so (git-ignored; travels to the GPU box).  Flags go to the lab kernel (e.g. -DXS_LAB=1: timing
# ablations, -DXS_NSETS=3, -DXS_SPREAD=0, -DPOPE_XSTAT_OVERLAP); NAME "product" builds an unmodified copy for reference.
set -e
cd "$(dirname "$0")/../pope_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../scripts/_lab
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    if [ "$name" = product ]; then cp libpope_hip.so ../../scripts/_lab/libpope_product.so; echo "copied the product library"; continue; fi
    T=$(mktemp -d)
    $CXX $flags -c ../../scripts/gemm_xstat_lab.hip -o $T/gemm_xstat.o
    # the dispatch hook lives only in this patched copy of the product's launcher file
    { echo 'struct GemmParams; bool pope_xstat_supported(const GemmParams&); int pope_launch_xstat(const GemmParams&, struct ihipStream_t*);';
      sed 's|^    switch (g.epilogue) {$|    if (pope_xstat_supported(g)) return pope_launch_xstat(g, stream);\n    switch (g.epilogue) {|' gemm_f16x3.hip; } > $T/gemm_f16x3_lab.hip
    grep -q pope_launch_xstat\(g $T/gemm_f16x3_lab.hip
    $CXX -I. -c $T/gemm_f16x3_lab.hip -o $T/gemm_f16x3.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/_lab/libpope_$name.so \
        $(ls *.o | grep -v "^gemm_f16x3.o") $T/gemm_xstat.o $T/gemm_f16x3.o
    rm -rf $T
    echo "built scripts/_lab/libpope_$name.so [$flags]"
done
