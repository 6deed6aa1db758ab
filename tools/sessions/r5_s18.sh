#!/bin/bash
# round 5, GPU session 18: how much of the fused DMA statement's leading s_nop 4 is visible (np1: s_nop 0, np0: none; timing only until the ISA scan covers the SGPR hazard)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s18; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 900 python tools/ab_stage.py $A/libfa_sp1.so $A/libfa_np1.so $A/libfa_np0.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,fp16 d128 4k causal,fp16 d64 16k" --stages fwd --rounds 9 > $O/dma_pad_ab.log 2>&1
grep -v amdgpu.ids $O/dma_pad_ab.log
