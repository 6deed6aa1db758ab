#!/bin/bash
# round 5, GPU session 2: head_dim-64 backward 16x16x32: where it wins (thresholds), register-held V k-steps, prefetch depth, DMA stagger
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s2; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 900 python tools/ab_stage.py $A/libfa_base.so $A/libfa_d64.so $A/libfa_d64v0.so $A/libfa_d64v1.so $A/libfa_d64pf2.so $A/libfa_d64nostag.so --only "d64" --stages dq,dkdv --rounds 5 > $O/d64_bwd_ab2.log 2>&1
grep -c . $O/d64_bwd_ab2.log
