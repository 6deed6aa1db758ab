#!/bin/bash
# round 5, GPU session 10: timing-only ablations for a split softmax (second 32-key chunk's exponentials moved from the softmax phase into the wave's own next matrix phase):
# s16 = the softmax phase skips them (results wrong), s32 = the matrix phase carries them as extra work (results right, computed twice), s48 = both (the proposal's timing)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s10; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_s16.so $A/libfa_s32.so $A/libfa_s48.so --only "c3 fp16,c5shard,c2 fp16,fp16 d128 4k causal,fp16 d64 16k" --stages fwd --rounds 7 > $O/split_softmax_ablation.log 2>&1
grep -v amdgpu.ids $O/split_softmax_ablation.log
