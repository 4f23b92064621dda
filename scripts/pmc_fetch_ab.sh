#!/bin/bash
# Dev: fabric traffic (FETCH_SIZE, WRITE_SIZE) of the 64-image forward for each library given (POPE_LIB_PATH), two rocprofv3 --pmc
# passes each (counters only), summary lines of the QKV / FC1 / LN-fused GEMM kernels.  Usage: bash scripts/pmc_fetch_ab.sh lib.so ...
set -e -o pipefail
R=$(pwd); export TMPDIR=/tmp
for lib in "$@"; do
  L=$(realpath $lib); name=$(basename $lib .so)
  for pass in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && POPE_LIB_PATH=$L timeout -k 10 420 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmcab_${name}_$pass -o pmc -- python3 $R/scripts/prof_forward.py 64 2 > $R/gpurun_out/pmcab.log 2>&1)
  done
  python3 scripts/pmc_summary.py $(find gpurun_out/pmcab_${name}_FETCH_SIZE gpurun_out/pmcab_${name}_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/pmcab_$name.txt
  rm -rf gpurun_out/pmcab_${name}_FETCH_SIZE gpurun_out/pmcab_${name}_WRITE_SIZE
  echo "== $name"
  python3 - gpurun_out/pmcab_$name.txt <<'PY'
import re, sys
cur = None
for line in open(sys.argv[1]):
    if not line.startswith(' '):
        cur = line.split('  (avg')[0].strip(); dur = re.search(r'avg dispatch ([\d.]+)', line).group(1); vals = {}
    else:
        m = re.match(r'\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)', line)
        if m: vals[m.group(1)] = float(m.group(2))
        if len(vals) == 2 and ('gemm_plain256' in cur or 'gemm_rowln16' in cur):
            print(f"  {cur[:56]:56s} {dur:>7s} us  fabric {(2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024 / 1e6:8.1f} MB (fetch x2 {2 * vals['FETCH_SIZE'] * 1024 / 1e6:7.1f})")
            vals = {}
PY
done
