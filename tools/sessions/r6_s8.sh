#!/bin/bash
# round 6, GPU session 8: end-of-round evidence (tools/round_evidence.sh) and the whole GPU suite on the same library, one lease
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/round_evidence.sh gpurun_out/evidence 2>&1 | tail -30
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/evidence/pytest_gpu.log 2>&1; tail -n 25 gpurun_out/evidence/pytest_gpu.log
