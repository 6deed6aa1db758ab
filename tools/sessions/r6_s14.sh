#!/bin/bash
# round 6, GPU session 14: the forward's role pieces issued between the MFMAs of the wave's matrix phase (FA_PP16_DMA_IN_M), alone and with deeper fragment prefetch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s14; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
for v in inm inmpf4; do timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_$v.so > $O/check_$v.log 2>&1; tail -n 2 $O/check_$v.log; timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_$v.so --d 64 > $O/check_${v}_d64.log 2>&1; tail -n 1 $O/check_${v}_d64.log; done
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_inm.so $A/libfa_inmpf3.so $A/libfa_inmpf4.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,fp16 d128 2k,fp16 d64 8k,bf16 d128 8k causal" --stages fwd --rounds 7 > $O/fwd_dma_in_m_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_dma_in_m_ab.log
for v in ftiminm ftiminmpf4; do timeout 300 python tools/phase_timing_fwd.py $A/libfa_$v.so > $O/phase_$v.log 2>&1; grep "group" $O/phase_$v.log; done
