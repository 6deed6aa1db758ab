#!/bin/bash
# round 5, GPU session 14: the ablation step of bench.py's roofline.ceiling chain on the shipped library (profiles/r5_fwd_ceiling_ablations.json), then bench.py so that the line shows it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s14; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python tools/fwd_ceiling_ablations.py $O/fwd_ceiling_ablations.json > $O/fwd_ceiling_ablations.log 2>&1; echo "ablations rc=$?"
tail -n 30 $O/fwd_ceiling_ablations.log
