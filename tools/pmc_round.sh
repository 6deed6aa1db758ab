#!/bin/bash
# Shader-side PMC passes for the round's evidence: forward at C3 (fp16 b4 s16384 causal) and backward at C4 (bf16 b4 s8192, non-causal and
# causal).  One counter group per pass, --kernel-trace only (never with sys / hip / hsa / memory-copy tracing).  tools/summarize_counters.py
# reduces the CSVs.  Usage (on the GPU box): bash tools/pmc_round.sh OUTDIR
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/pmc_round}
case "$OUT" in /*) ;; *) OUT="$(pwd)/$OUT";; esac
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
GROUPS_=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS")
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/fwdc3_g${i}" -- python "$R/tools/run_fwd_once.py" --seq 16384 --causal 1 --iters 4 > "$OUT/fwdc3_g${i}.stdout" 2>&1
  echo "fwd c3 group $i rc=$?"
  for causal in 0 1; do
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/bwdc4_causal${causal}_g${i}" -- python "$R/tools/run_bwd_once.py" 4 8192 32 32 128 bf16 $causal 4 > "$OUT/bwdc4_causal${causal}_g${i}.stdout" 2>&1
    echo "bwd c4 causal=$causal group $i rc=$?"
  done
done
