#!/bin/bash
# round 5, GPU session 5: forward with three query columns per wave (384-row workgroups, 32-key tiles): values first, then A/B against the product and the
# 2-column / 32-key build that isolates the shorter phases
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s5; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_qb3.so > $O/check_qb3.log 2>&1; echo "check qb3 rc=$?"
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_qb2bn32.so > $O/check_qb2bn32.log 2>&1; echo "check qb2bn32 rc=$?"
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_base.so > $O/check_base.log 2>&1; echo "check base rc=$?"
tail -4 $O/check_qb3.log
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_qb3.so $A/libfa_qb2bn32.so --only "c3 fp16,c5shard,c2 fp16,c4 bf16,bf16 d128 8k causal,fp16 d128 2k,fp16 d128 4k causal,fp16 d128 1k" --stages fwd --rounds 7 > $O/qb3_ab.log 2>&1
cat $O/qb3_ab.log
