#!/bin/bash
# round 5, GPU session 12: v_pk_fma_f32 for the score scaling in the 16x16x32 forward (head_dim 128 instances only: the head_dim-64 ones spill with it)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s12; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_pk1.so --policy 1 --d 128 > $O/check_pk1_d128.log 2>&1; echo "check pk1 d128 rc=$?"
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_pk1.so --only "c3 fp16,c5shard,c2 fp16,fp16 d128 2k,fp16 d128 4k causal,c4 bf16,bf16 d128 8k causal" --stages fwd --rounds 7 > $O/pk_fma_ab.log 2>&1
grep -v amdgpu.ids $O/pk_fma_ab.log | grep "B:pk1"
