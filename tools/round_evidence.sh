#!/bin/bash
# End-of-round evidence, ONE box, one lease (run on the GPU box: tools/gpu.sh 2400 'bash tools/round_evidence.sh'):
#   1. plain `python bench.py` (the driver's command)                                   -> bench_stdout.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command, headline (c3) and c4       -> bench_stats/, bench_c4_stats/
#   3. HBM counters, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel-trace only), forward at C3 and backward at C4
#   4. shader counters (four groups, separate passes), same two cases
# tools/round_evidence_collect.py then writes profiles/rNN_* with the library's source digest and the git commit in every summary.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/evidence}
case "$OUT" in /*) ;; *) OUT="$(pwd)/$OUT";; esac
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
timeout 900 python "$R/bench.py" > "$OUT/bench_stdout.json" 2> "$OUT/bench_stderr.log"; echo "bench rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python "$R/bench.py" --no-cpu-baseline --no-extra > "$OUT/bench_stdout_under_rocprof.json" 2> "$OUT/bench_rocprof_stderr.log"; echo "bench stats rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_c4_stats" -- python "$R/bench.py" --workload c4 --no-cpu-baseline --no-extra > "$OUT/bench_c4_stdout_under_rocprof.json" 2> "$OUT/bench_c4_rocprof_stderr.log"; echo "bench c4 stats rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/fwd_c3_$ctr" -- python "$R/tools/run_fwd_once.py" --seq 16384 --causal 1 --iters 4 > "$OUT/fwd_c3_$ctr.stdout" 2>&1; echo "fwd c3 $ctr rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/bwd_c4_$ctr" -- python "$R/tools/run_bwd_once.py" 4 8192 32 32 128 bf16 0 4 > "$OUT/bwd_c4_$ctr.stdout" 2>&1; echo "bwd c4 $ctr rc=$?"
done
GROUPS_=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU")
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/sq/fwdc3_g${i}" -- python "$R/tools/run_fwd_once.py" --seq 16384 --causal 1 --iters 4 > "$OUT/sq_fwdc3_g${i}.stdout" 2>&1; echo "sq fwd c3 group $i rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/sq/bwdc4_g${i}" -- python "$R/tools/run_bwd_once.py" 4 8192 32 32 128 bf16 0 4 > "$OUT/sq_bwdc4_g${i}.stdout" 2>&1; echo "sq bwd c4 group $i rc=$?"
done
find "$OUT" -name "*kernel_trace.csv" -size +6M -delete      # per-dispatch traces can be large; the stats / counter CSVs are small
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
