#!/bin/bash
# round 5, GPU session 1: d64 backward 16x16x32 (two tile widths), bf16 MFMA row sums, DMA-protocol debug builds, fixed-cost fit
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s1; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_d64bwd.so $A/libfa_d64bwdbn128.so --only "d64 8k,d64 2k,d64 4k,d64 16k" --stages dq,dkdv --rounds 5 > $O/d64_bwd_ab.log 2>&1
timeout 300 python tools/ab_stage.py $A/libfa_base.so $A/libfa_bf16ml.so --only "bf16 d128,bf16 d64" --stages fwd --rounds 5 > $O/bf16_ml_ab.log 2>&1
FA_GFX950_LIBRARY=$A/libfa_bf16ml.so timeout 600 python tools/lse_error.py --only bf16 --out $O/lse_error_bf16ml.json > $O/lse_error_bf16ml.log 2>&1
timeout 400 python tools/ab_stage.py $A/libfa_base.so $A/libfa_late.so $A/libfa_sleepy.so $A/libfa_racy.so $A/libfa_racylate.so --only "c2 fp16,bf16 d128 8k causal,fp16 d128 3k causal,c4 bf16" --stages fwd --rounds 2 --iters 2 > $O/dma_debug_fwd.log 2>&1
timeout 400 python tools/ab_stage.py $A/libfa_base.so $A/libfa_late.so $A/libfa_sleepy.so --only "c4 bf16,bf16 d128 8k causal,fp16 d128 3k causal,gqa 4k" --stages dkdv --rounds 2 --iters 2 > $O/dma_debug_dkdv.log 2>&1
timeout 300 python tools/fixed_cost.py > $O/fixed_cost.log 2>&1
timeout 200 tools/clockbench > $O/clockbench.log 2>&1
tail -n 40 $O/d64_bwd_ab.log
