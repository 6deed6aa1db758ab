#!/bin/bash
# round 6, GPU session 11: the forward's MFMA-summed tiles with their row sums by v_dot2c_f32_f16 in the softmax phase instead of four MFMAs per tile
# (under the power cap an MFMA costs more energy than the 16 VALU that replace four of them)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s11; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_dot2.so > $O/check_dot2.log 2>&1; tail -n 4 $O/check_dot2.log
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_dot2.so --d 64 > $O/check_dot2_d64.log 2>&1; tail -n 4 $O/check_dot2_d64.log
timeout 700 python tools/ab_stage.py $A/libfa_base.so $A/libfa_dot2.so --only "c3 fp16,c5shard,c2 fp16,fp16 d128 4k causal,fp16 d128 2k,fp16 d128 8k,fp16 d64 8k,fp16 d64 8k causal" --stages fwd --rounds 9 > $O/fwd_rowsum_dot2_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_rowsum_dot2_ab.log
