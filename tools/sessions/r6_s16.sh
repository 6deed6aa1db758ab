#!/bin/bash
# round 6, GPU session 16: issue priority 2 (1) for a wave's softmax side in the forward's steady loop (FA_PP16_S_PRIO) - head_dim 64's softmax side is the critical one
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s16; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_sprio.so $A/libfa_sprio1.so --only "c3 fp16,c5shard,c2 fp16,fp16 d64 8k,fp16 d64 2k,fp16 d64 4k,fp16 d64 16k" --stages fwd --rounds 7 > $O/fwd_softmax_prio_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_softmax_prio_ab.log
timeout 200 python tools/phase_timing_fwd.py $A/libfa_ftimsprio.so > $O/phase_ftimsprio.log 2>&1; grep -E "group" $O/phase_ftimsprio.log
