#!/bin/bash
# round 6, GPU session 10: the SIMPLE forward instance with every other workgroup of a compute unit starting ~4 us late
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s10; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_fsimple.so $A/libfa_fsimst1.so $A/libfa_fsimst32.so --only "fp16 d128 512,fp16 d128 1k,fp16 d128 2k" --stages fwd --rounds 9 > $O/fwd_simple_stagger_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_simple_stagger_ab.log
