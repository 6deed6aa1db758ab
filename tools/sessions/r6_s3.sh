#!/bin/bash
# round 6, GPU session 3: the launch-size-aware policy on the small grids (after), bench.py with the new ceiling chain, the whole GPU suite with the four-rule tolerance bookkeeping
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s3; mkdir -p $O
export PYTHONUNBUFFERED=1
for g in "1 8" "1 32" "2 16" "4 32"; do set -- $g; timeout 400 python tools/ab_policy_sweep.py --b $1 --h $2 --d 128 --dtype fp16 --seqs 512,1024,2048,4096,8192 >> $O/policy_small_grids.log 2>&1; done
grep "^#" $O/policy_small_grids.log
timeout 900 python bench.py > $O/bench_stdout.json 2> $O/bench_stderr.log; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6s3/bench_stdout.json") if l.startswith("{")][-1])
r=d["roofline"]; print({k:v for k,v in r.items() if not isinstance(v,(dict,list))})
print(json.dumps(r["ceiling"]["chain_tflops"],indent=1)); print(d["extra"].get("n1_consistency")); print(d["value"], d.get("tflops_at_median_launch"))
print(d["extra"].get("d64_b4_s8192_h32_fp16"))
PY
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -n 15 $O/pytest_gpu.log
