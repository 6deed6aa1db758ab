#!/bin/bash
# round 6, GPU session 15: the forward's steady loop with ONE workgroup barrier per step (FA_PP16_ONE_BARRIER)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s15; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 200 python tools/check_variant_fwd.py --lib $A/libfa_oneb.so > $O/check_oneb.log 2>&1; tail -n 3 $O/check_oneb.log
timeout 200 python tools/check_variant_fwd.py --lib $A/libfa_oneb.so --d 64 > $O/check_oneb_d64.log 2>&1; tail -n 2 $O/check_oneb_d64.log
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_oneb.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,fp16 d128 2k,fp16 d128 1k,fp16 d64 8k,bf16 d128 8k causal,fp16 d64 2k" --stages fwd --rounds 9 > $O/fwd_one_barrier_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_one_barrier_ab.log
timeout 200 python tools/phase_timing_fwd.py $A/libfa_ftimoneb.so > $O/phase_ftimoneb.log 2>&1; grep -E "group|per wave" $O/phase_ftimoneb.log
