#!/bin/bash
# round 5, GPU session 15: determinism soak, per-kernel sweep over the reference's benchmark grid (+ charts), efficiency sweep - all on the shipped library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s15; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python tools/soak.py --iters 150 > $O/soak.log 2>&1; echo "soak rc=$?"; tail -n 3 $O/soak.log
timeout 1200 python tools/sweep_profile.py --out $O/sweep_profile.csv > $O/sweep_profile.log 2>&1; echo "sweep_profile rc=$?"

timeout 900 python tools/sweep_efficiency.py > $O/sweep_efficiency.log 2>&1; echo "sweep_efficiency rc=$?"; tail -n 5 $O/sweep_efficiency.log
