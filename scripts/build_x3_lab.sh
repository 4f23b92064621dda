#!/bin/bash
# Dev: build scripts/x3_lab (GEMM lab harness); pass extra -D flags, e.g. -DX3_STAMPS, for experiment variants.
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude"
hipcc $FLAGS "$@" -c pope_amd/csrc/gemm_f16x3.hip -o /tmp/x3_gemm_f16x3.o
hipcc $FLAGS "$@" -x hip -c scripts/x3_lab.cpp -o /tmp/x3_lab.o
hipcc --offload-arch=gfx950 -o scripts/x3_lab /tmp/x3_lab.o pope_amd/csrc/gemm_f32.o /tmp/x3_gemm_f16x3.o pope_amd/csrc/layernorm.o
