#!/bin/bash
# round 5, GPU session 13: final library - full -m gpu suite, the driver's bench command line, then the end-of-round evidence set
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s13; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.log 2>&1
echo rc=$? >> $O/pytest_gpu.log
tail -n 4 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra > $O/bench_driver_cmdline.json 2> $O/bench_driver_cmdline.err; echo "driver-style bench rc=$?"
python -c "import json,sys; d=json.loads([l for l in open('$O/bench_driver_cmdline.json') if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
bash tools/round_evidence.sh
