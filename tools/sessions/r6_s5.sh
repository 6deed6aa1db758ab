#!/bin/bash
# round 6, GPU session 5: the request / statistics duties of dK/dV moved to waves 4-7 (FA_KV16_Q1_DUTIES), the skewed dQ schedule (FA_DQ16_SKEW), the N = 1 consistency check, the packed grid with threaded oracle jobs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s5; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_kvq1.so $A/libfa_kvq2.so --only "c4 bf16,bf16 d128 8k causal,c2 fp16,fp16 d128 2k,c3 fp16,gqa 4k causal" --stages dkdv --rounds 7 > $O/dkdv_q1_duties_ab.log 2>&1
grep -v amdgpu.ids $O/dkdv_q1_duties_ab.log
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_dqskew.so --only "c4 bf16,bf16 d128 8k causal,c2 fp16,fp16 d128 2k,c3 fp16,fp16 d64 8k,fp16 d64 8k causal,fp16 d128 4k causal" --stages dq --policy 1 --rounds 7 > $O/dq_skew_ab.log 2>&1
grep -v amdgpu.ids $O/dq_skew_ab.log
( time timeout 900 python -m pytest tests/test_perf_relations_gpu.py tests/test_attention_gpu.py -m gpu -q -x -k "n1_row or varlen_grid" --durations=5 ) > $O/pytest_part.log 2>&1; tail -n 25 $O/pytest_part.log
