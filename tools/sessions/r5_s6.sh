#!/bin/bash
# round 5, GPU session 6: which keys do the 32-key-tile experiment builds get wrong under a causal mask?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s6; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 200 python tools/debug_keys.py --lib $A/libfa_qb3.so --sq 512 --sk 512 > $O/keys_qb3_512.log 2>&1
timeout 200 python tools/debug_keys.py --lib $A/libfa_qb3.so --sq 2048 --sk 2048 --max-rows 60 > $O/keys_qb3_2048.log 2>&1
timeout 200 python tools/debug_keys.py --lib $A/libfa_qb2bn32.so --sq 4096 --sk 4100 --dtype bf16 --max-rows 60 > $O/keys_qb2bn32_bf16.log 2>&1
timeout 200 python tools/debug_keys.py --lib $A/libfa_base.so --sq 2048 --sk 2048 > $O/keys_base.log 2>&1
head -50 $O/keys_qb3_512.log
