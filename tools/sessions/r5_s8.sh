#!/bin/bash
# round 5, GPU session 8: folded row assignment under a causal mask (forward 16x16x32): values, then A/B; dK/dV with its LDS-DMA pieces spread between MFMAs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s8; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_fold.so > $O/check_fold_d128.log 2>&1; echo "check fold d128 rc=$?"
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_fold.so --d 64 > $O/check_fold_d64.log 2>&1; echo "check fold d64 rc=$?"
timeout 900 python tools/ab_stage.py $A/libfa_nofold.so $A/libfa_fold.so --policy 1 --only "causal" --stages fwd --rounds 7 > $O/fold_ab_pinned16.log 2>&1
timeout 600 python tools/ab_stage.py $A/libfa_nofold.so $A/libfa_fold.so --only "causal" --stages fwd --rounds 7 > $O/fold_ab_auto.log 2>&1
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_kvs1.so $A/libfa_kvs2.so --only "c4 bf16,bf16 d128 8k causal,c3 fp16,c2 fp16,fp16 d128 2k,fp16 d64 8k,bf16 d128 8k gqa4,fp16 d128 gqa 4k" --stages dkdv --rounds 7 > $O/kv_spread_ab.log 2>&1
grep -v amdgpu.ids $O/fold_ab_pinned16.log | grep "B:fold"
grep -v amdgpu.ids $O/kv_spread_ab.log | grep -v "A:base"
timeout 900 python -m pytest tests/test_attention_gpu.py -m gpu -q -k "test_reference_varlen_grid_vs_torch_fp32" > $O/pytest_varlen_grid.log 2>&1; echo "varlen grid rc=$?"; tail -n 3 $O/pytest_varlen_grid.log
