#!/bin/bash
# round 5, GPU session 17: fused LDS-DMA statements in the backward 16x16x32 kernels (two pieces per statement) on top of the forward's (four pieces + skipped pad): new against old
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s17; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 1200 python tools/ab_stage.py $A/libfa_old.so $A/libfa_new.so --only "c4 bf16,bf16 d128 8k causal,c3 fp16,c2 fp16,fp16 d128 2k,fp16 d64 8k,fp16 d64 16k,bf16 d128 8k gqa4,fp16 d128 gqa 4k,bf16 d128 8k mqa causal" --stages fwd,dq,dkdv --rounds 7 > $O/dma_fused_all_ab.log 2>&1
grep -v amdgpu.ids $O/dma_fused_all_ab.log | grep "B:new"
