#!/bin/bash
# round 5, GPU session 26: -m * scale cached next to the running max (no multiply at the top of the softmax phase): mc1 against the product
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s26; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_mc1.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,fp16 d128 2k,fp16 d128 4k causal,bf16 d128 8k causal,fp16 d64 16k,fp16 d64 8k" --stages fwd --rounds 9 > $O/mc_cache_ab.log 2>&1
grep -v amdgpu.ids $O/mc_cache_ab.log | grep "B:mc1"
