#!/bin/bash
# round 5, GPU session 20: the 32x32x16 forward with a wave's two K (V) pieces per tile as one asm statement (ppf1) against per-piece statements (ppf0); pinned to the 32x32x16 set
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s20; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_ppf1.so --policy 0 --d 128 > $O/check_ppf1_d128.log 2>&1; echo "check ppf1 d128 rc=$?"
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_ppf1.so --policy 0 --d 64 > $O/check_ppf1_d64.log 2>&1; echo "check ppf1 d64 rc=$?"
timeout 900 python tools/ab_stage.py $A/libfa_ppf0.so $A/libfa_ppf1.so --policy 0 --only "fp16 d128 512,fp16 d128 1k,fp16 d128 2k,c2 fp16,fp16 d128 4k causal,c3 fp16,c4 bf16,bf16 d64 8k,fp16 d64 16k,fp16 d64 2k,fp16 d64 1k,bf16 d128 2k" --stages fwd --rounds 9 --iters 5 > $O/pp_dma_fused_ab.log 2>&1
grep -v amdgpu.ids $O/pp_dma_fused_ab.log | grep "B:ppf1"
