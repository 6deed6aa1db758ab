#!/bin/bash
# round 6, GPU session 7: both head dims over launch sizes on the final round-6 policy
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s7; mkdir -p $O
export PYTHONUNBUFFERED=1
for g in "4 32" "1 32" "1 8" "16 32"; do set -- $g; S=512,1024,2048,4096,8192,16384; [ "$1" = 16 ] && S=512,1024,2048,4096; timeout 600 python tools/ab_policy_sweep.py --b $1 --h $2 --d 64 --dtype fp16 --seqs $S >> $O/policy_d64.log 2>&1; done
grep "(!)\|^#" $O/policy_d64.log | cut -c1-420
for g in "1 8" "1 32" "2 16" "4 32"; do set -- $g; timeout 400 python tools/ab_policy_sweep.py --b $1 --h $2 --d 128 --dtype fp16 --seqs 512,1024,2048,4096,8192 >> $O/policy_small_grids.log 2>&1; done
grep "(!)\|^#" $O/policy_small_grids.log | cut -c1-420
