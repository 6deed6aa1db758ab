#!/bin/bash
# round 6, GPU session 13: the forward's LDS-DMA pieces spread over the softmax pass (branch-free), alone and with deeper fragment prefetch in the matrix phase
# (the stamps of session 12 say the two sides of a step are balanced, 1310 vs 1330 cycles: a gain on one side alone cannot show)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s13; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
for v in spread spreadpf4; do timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_$v.so > $O/check_$v.log 2>&1; tail -n 2 $O/check_$v.log; done
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_spread.so $A/libfa_spreadpf3.so $A/libfa_spreadpf4.so $A/libfa_pf3.so $A/libfa_pf4.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,fp16 d128 2k,fp16 d64 8k" --stages fwd --rounds 7 > $O/fwd_dma_spread_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_dma_spread_ab.log
for v in ftimspread ftimspreadpf4; do timeout 300 python tools/phase_timing_fwd.py $A/libfa_$v.so > $O/phase_$v.log 2>&1; grep "group" $O/phase_$v.log; done
