#!/bin/bash
# round 5, GPU session 23: the steady loop's LDS-DMA requests issued BEFORE the barrier that ends the matrix phase (early) against the product (behind it)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s23; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_early.so --policy 1 --d 128 > $O/check_early_d128.log 2>&1; echo "check early d128 rc=$?"
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_early.so --policy 1 --d 64 > $O/check_early_d64.log 2>&1; echo "check early d64 rc=$?"
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_early.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,fp16 d128 2k,fp16 d128 4k causal,bf16 d128 8k causal,fp16 d64 16k,fp16 d64 8k" --stages fwd --rounds 9 > $O/dma_early_ab.log 2>&1
grep -v amdgpu.ids $O/dma_early_ab.log | grep "B:early"
