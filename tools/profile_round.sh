#!/bin/bash
# Per-round profiler evidence (run on the GPU box): (1) rocprofv3 --kernel-trace --stats of the default bench command,
# (2) HBM-side counters of the forward at C3 and of the backward at C4, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (kernel-trace only, no sys/hip trace).  tools/summarize_pmc.py turns the CSVs into profiles/rNN_*.json.
# Usage: tools/profile_round.sh OUTDIR
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/profile_round}
case "$OUT" in /*) ;; *) OUT="$(pwd)/$OUT";; esac
export TMPDIR=/tmp
mkdir -p "$OUT"; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extra > "$OUT/bench_stdout.json" 2> "$OUT/bench_stderr.log"
echo "bench stats rc=$?"
# the same for BASELINE configs[3] (forward + backward, b4 s8192 bf16): per-kernel averages of dot_do_o, dQ, dK/dV next to the forward
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_c4_stats" -- python "$R/bench.py" --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > "$OUT/bench_c4_stdout.json" 2> "$OUT/bench_c4_stderr.log"
echo "bench c4 stats rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/fwd_c3_$ctr" -- python "$R/tools/run_fwd_once.py" --seq 16384 --causal 1 --iters 4 > "$OUT/fwd_c3_$ctr.stdout" 2>&1
  echo "fwd c3 $ctr rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/bwd_c4_$ctr" -- python "$R/tools/run_bwd_once.py" 4 8192 32 32 128 bf16 0 4 > "$OUT/bwd_c4_$ctr.stdout" 2>&1
  echo "bwd c4 $ctr rc=$?"
done
find "$OUT" -name "*.csv" -size +8M -delete      # per-dispatch traces of the bench run can be large; the stats CSVs are small
