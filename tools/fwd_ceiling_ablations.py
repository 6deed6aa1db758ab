#!/usr/bin/env python3
"""profiles/rNN_fwd_ceiling_ablations.json for bench.py's `roofline.ceiling`: time ratios of timing-only ablation builds of the 16x16x32 forward (results WRONG by
construction) against the shipped kernel, interleaved in one process on one box.  Builds (tools/build_variant.py): base; abl8 = -DFA_PP16_ABL=8 (no LDS fragment reads in
the matrix phases); abl4 = 4 (no exponentials: one multiply per score); abl2 = 2 (no LDS-DMA issued in the steady loop); abl14 = 14 (all three).
Usage: fwd_ceiling_ablations.py OUT.json [--rounds 7]"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
NAMES = {"abl8": "no_lds_fragment_reads", "abl4": "no_exponentials", "abl2": "no_lds_dma", "abl14": "no_lds_reads_no_exp_no_dma"}
WORK = {"c3": "c3 fp16", "c5shard": "c5shard", "c2": "c2 fp16", "c4": "c4 bf16"}

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
libs = [os.path.join(ROOT, "tools", "abl", f"libfa_{n}.so") for n in ["base"] + list(NAMES)]
cmd = [sys.executable, os.path.join(ROOT, "tools", "ab_stage.py")] + libs + ["--policy", "1", "--only", ",".join(WORK.values()), "--stages", "fwd", "--rounds", str(a.rounds)]
text = subprocess.run(cmd, capture_output=True, text=True, timeout=1500).stdout
from flash_attn_turing import capi  # noqa: E402

out = {"what": "time of an ablation build / time of the shipped kernel, forward, kernel policy pinned to the 16x16x32 set, median of interleaved rounds (tools/ab_stage.py); "
               "the ablated builds compute WRONG results and exist only to price what they leave out",
       "library": capi.lib().fa_build_info().decode(), "library_source_digest": re.search(r"src=(\w+)", capi.lib().fa_build_info().decode()).group(1),
       "workloads": {}, "raw": [ln for ln in text.splitlines() if " fwd " in ln]}
for key, pat in WORK.items():
    rows = [ln for ln in text.splitlines() if ln.startswith(pat) and " fwd " in ln]
    ratios, tf = {}, {}
    for ln in rows:
        m = re.search(r"fwd\s+\w:(\w+)\s+([\d.]+) ms .*?\s(\d+) TF\s+vs A\s+([\d.]+)", ln)
        if not m:
            continue
        tf[m.group(1)] = int(m.group(3))
        if m.group(1) in NAMES:
            ratios[NAMES[m.group(1)]] = float(m.group(4))
    if ratios:
        out["workloads"][key] = {"shipped_tflops": tf.get("base"), "time_ratio_vs_shipped": ratios, "tflops": {NAMES[k]: v for k, v in tf.items() if k in NAMES}}
with open(a.out, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["workloads"], indent=1))
