#!/bin/bash
# round 5, GPU session 19: LDS-DMA requests BEHIND the softmax (no pad: scores read in the order their MFMAs wrote them) - dl1 against sp1 = product
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s19; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_dl1.so --policy 1 --d 128 > $O/check_dl1_d128.log 2>&1; echo "check dl1 d128 rc=$?"
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_dl1.so --policy 1 --d 64 > $O/check_dl1_d64.log 2>&1; echo "check dl1 d64 rc=$?"
timeout 900 python tools/ab_stage.py $A/libfa_sp1.so $A/libfa_dl1.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,fp16 d128 2k,fp16 d128 4k causal,bf16 d128 8k causal,fp16 d64 16k,fp16 d64 8k,sq16k sk2k" --stages fwd --rounds 9 > $O/dma_last_ab.log 2>&1
grep -v amdgpu.ids $O/dma_last_ab.log | grep "B:dl1"
