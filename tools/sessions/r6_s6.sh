#!/bin/bash
# round 6, GPU session 6: the two kernel sets at head_dim 64 (forward fp16, dQ, dK/dV) over launch sizes - where does FA_POLICY_AUTO stand
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s6; mkdir -p $O
export PYTHONUNBUFFERED=1
for g in "4 32" "1 32" "1 8" "16 32"; do set -- $g; S=512,1024,2048,4096,8192,16384; [ "$1" = 16 ] && S=512,1024,2048,4096; timeout 600 python tools/ab_policy_sweep.py --b $1 --h $2 --d 64 --dtype fp16 --seqs $S >> $O/policy_d64.log 2>&1; done
grep -v amdgpu.ids $O/policy_d64.log
