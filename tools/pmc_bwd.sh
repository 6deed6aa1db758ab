#!/bin/bash
# Backward-kernel PMC passes at C4 (bf16 b4 s8192 h32 d128), non-causal and causal.  Counter passes are run on their own with only
# --kernel-trace (never together with sys / hip / hsa / memory-copy tracing).  Usage (on the GPU box): bash tools/pmc_bwd.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/pmc_bwd}
case "$OUT" in /*) ;; *) OUT="$(pwd)/$OUT";; esac     # absolute: the passes run from /tmp
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
for causal in 0 1; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/c${causal}_g${i}" -- python "$R/tools/run_bwd_once.py" 4 8192 32 32 128 bf16 $causal 3 > "$OUT/c${causal}_g${i}.stdout" 2> "$OUT/c${causal}_g${i}.stderr"
    echo "causal=$causal group $i rc=$? : $(find "$OUT/c${causal}_g${i}" -name '*counter_collection.csv' | head -1)"
  done
done
