#!/bin/bash
# round 6, GPU session 4: the skewed dK/dV schedule (FA_KV16_SKEW) against the lock-step product, three request placements and without the static priority offset
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s4; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_kvskew.so $A/libfa_kvskew1.so $A/libfa_kvskew2.so $A/libfa_kvskewp0.so --only "c4 bf16,bf16 d128 8k causal,c2 fp16,fp16 d128 2k,fp16 d128 4k causal,c3 fp16,fp16 d128 1k,gqa 4k causal,mqa causal" --stages dkdv --rounds 7 > $O/dkdv_skew_ab.log 2>&1
grep -v amdgpu.ids $O/dkdv_skew_ab.log
( time timeout 2400 python -m pytest tests -m gpu -q --durations=25 ) > $O/pytest_gpu.log 2>&1; tail -n 45 $O/pytest_gpu.log
