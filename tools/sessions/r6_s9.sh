#!/bin/bash
# round 6, GPU session 9: the SIMPLE forward instance (four waves, 128 rows, 64 KiB of LDS, two workgroups per compute unit) on short sequences
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s9; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_fsimple.so --policy 2 > $O/check_fsimple.log 2>&1; tail -n 22 $O/check_fsimple.log
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_fsimple.so --only "fp16 d128 512,fp16 d128 1k,fp16 d128 2k,c2 fp16,fp16 d128 4k causal,fp16 d128 3k causal,bf16 d128 2k mqa" --stages fwd --rounds 9 > $O/fwd_simple_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_simple_ab.log
