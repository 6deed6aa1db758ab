#!/bin/bash
# round 5, GPU session 22: dQ 16x16x32 with all of S before all of dP in a 32-key half (the exponentials of S under the dP MFMAs): dqsf against the product
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s22; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_dqsf.so --policy 1 --only "c4 bf16,bf16 d128 8k causal,c3 fp16,c2 fp16,fp16 d128 2k,fp16 d64 8k,fp16 d64 16k,bf16 d128 8k gqa4" --stages dq --rounds 7 > $O/dq_s_first_ab.log 2>&1
grep -v amdgpu.ids $O/dq_s_first_ab.log | grep "B:dqsf"
