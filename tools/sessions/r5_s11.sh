#!/bin/bash
# round 5, GPU session 11: row-sum MFMAs issued at the end of the softmax phase (rs1) against riding in the next matrix phase (rs0 = the product so far)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s11; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_rs1.so --policy 1 --d 128 > $O/check_rs1_d128.log 2>&1; echo "check rs1 d128 rc=$?"
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_rs1.so --policy 1 --d 64 > $O/check_rs1_d64.log 2>&1; echo "check rs1 d64 rc=$?"
timeout 900 python tools/ab_stage.py $A/libfa_rs0.so $A/libfa_rs1.so --only "c3 fp16,c5shard,c2 fp16,fp16 d128 2k,fp16 d128 4k causal,fp16 d64 16k,fp16 d64 8k,fp16 d64 4k,fp16 d128 gqa 4k,sq16k sk2k" --stages fwd --rounds 7 > $O/rowsum_in_s_ab.log 2>&1
grep -v amdgpu.ids $O/rowsum_in_s_ab.log | grep "B:rs1"
