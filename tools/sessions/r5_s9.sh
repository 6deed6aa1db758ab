#!/bin/bash
# round 5, GPU session 9: forward prologue that retires Q and K(0) by count (V(0) / K(1) stay in flight): values under both kernel sets and head dims, then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s9; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
for pol in 0 1; do for d in 128 64; do
  timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_pcnt.so --policy $pol --d $d > $O/check_pcnt_p${pol}_d$d.log 2>&1; echo "check pcnt policy $pol d $d rc=$?"
done; done
timeout 900 python tools/ab_stage.py $A/libfa_pzero.so $A/libfa_pcnt.so --only "fp16 d128 512,fp16 d128 1k,fp16 d128 2k,c2 fp16,fp16 d128 4k causal,fp16 d64 512,fp16 d64 1k,fp16 d64 2k,bf16 d128 2k,c3 fp16,c4 bf16,sq8k sk1k" --stages fwd --rounds 9 --iters 10 > $O/prologue_ab.log 2>&1
grep -v amdgpu.ids $O/prologue_ab.log | grep "B:pcnt"
