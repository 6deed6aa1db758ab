#!/bin/bash
# HBM-side traffic of the backward kernels at C4: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/pmc_bwd_hbm}
case "$OUT" in /*) ;; *) OUT="$(pwd)/$OUT";; esac
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
for causal in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/c${causal}_$ctr" -- python "$R/tools/run_bwd_once.py" 4 8192 32 32 128 bf16 $causal 3 > "$OUT/c${causal}_$ctr.stdout" 2> "$OUT/c${causal}_$ctr.stderr"
    echo "causal=$causal $ctr rc=$?"
  done
done
