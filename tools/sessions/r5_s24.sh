#!/bin/bash
# round 5, GPU session 24: where the two forward kernel sets cross after the softmax-phase trim (AUTO thresholds: 2^22 pairs per head, 2^24 under a mask; head_dim 64 fp16: 2^24 / 2^26)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s24; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python tools/ab_policy_sweep.py --seqs 512,1024,1536,2048,3072,4096,8192 --rounds 7 > $O/policy_d128_fp16.log 2>&1
timeout 600 python tools/ab_policy_sweep.py --dtype bf16 --seqs 1024,2048,4096,8192 --rounds 7 > $O/policy_d128_bf16.log 2>&1
timeout 600 python tools/ab_policy_sweep.py --d 64 --seqs 2048,4096,8192,16384 --rounds 7 > $O/policy_d64_fp16.log 2>&1
grep -v amdgpu.ids $O/policy_d128_fp16.log | cut -c1-95
grep -v amdgpu.ids $O/policy_d128_bf16.log | cut -c1-95
grep -v amdgpu.ids $O/policy_d64_fp16.log | cut -c1-95
