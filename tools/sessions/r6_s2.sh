#!/bin/bash
# round 6, GPU session 2: packed multiply-subtract in the forward's softmax (v_pk_fma_f32), VALU streams beside an MFMA partner, the small-grid policy sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s2; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 120 tools/ubench > $O/ubench.log 2>&1
sed -n '/round 6/,/priorities/p' $O/ubench.log
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_fpk2.so > $O/check_fpk2.log 2>&1; tail -n 3 $O/check_fpk2.log
timeout 600 python tools/ab_stage.py $A/libfa_base.so $A/libfa_fpk2.so --only "c3 fp16,c5shard,c2 fp16,fp16 d128 4k causal,c4 bf16,fp16 d128 2k,bf16 d128 8k causal,fp16 d64 8k,fp16 d64 16k" --stages fwd --rounds 9 > $O/fwd_pk_fma_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_pk_fma_ab.log
for g in "1 8" "1 32" "2 16"; do set -- $g; timeout 400 python tools/ab_policy_sweep.py --b $1 --h $2 --d 128 --dtype fp16 --seqs 512,1024,2048,4096,8192 >> $O/policy_small_grids.log 2>&1; done
grep -v amdgpu.ids $O/policy_small_grids.log
