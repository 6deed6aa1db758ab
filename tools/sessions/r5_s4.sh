#!/bin/bash
# round 5, GPU session 4 (re-entry after the container was re-created): full -m gpu suite with durations, then the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s4; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --durations=60 -x > $O/pytest_gpu.log 2>&1
echo rc=$? >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_stdout.json 2> $O/bench_stderr.log; echo bench rc=$?
tail -c 1500 $O/bench_stdout.json
