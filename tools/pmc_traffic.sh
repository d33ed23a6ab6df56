#!/bin/bash
# HBM-side traffic of the Linear GEMM family, one shape at a time (bench.py itself crashes under --pmc in this ROCm
# build).  Run on the GPU box from the repo root; writes gpurun_out/pmc/<shape>_<counter>/ then tools/pmc_traffic.py
# turns the CSVs into profiles/rNN_traffic.json.  FETCH_SIZE and WRITE_SIZE are collected in separate passes.
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
while read -r M K N G; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=$root/gpurun_out/pmc/${M}x${K}x${N}${G:+g}_$c
    mkdir -p "$out"
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out" -- python3 "$root/tools/one_gemm.py" $M $K $N $G > "$out/log.txt" 2>&1
    find "$out" -name "*kernel_trace.csv" -delete
  done
done <<'SHAPES'
1024 1280 1280
1024 5120 1280
1024 1280 3840
1024 1280 5120 g
4096 640 640
4096 640 1920
4096 640 2560 g
4096 2560 640
SHAPES
echo done
