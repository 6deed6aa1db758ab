#!/bin/bash
# round 5, GPU session 7: full -m gpu suite (durations) on the library with the tied softmax pad
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s7; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests -m gpu -q --durations=60 > $O/pytest_gpu.log 2>&1
echo rc=$? >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
