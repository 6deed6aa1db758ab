#!/bin/bash
# round 5, GPU session 3: full -m gpu suite with per-test durations (budgeting the new varlen grid) on the library with the head_dim-64 16x16x32 backward
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s3; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests -m gpu -q --durations=80 -x > $O/pytest_gpu.log 2>&1
echo rc=$? >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
