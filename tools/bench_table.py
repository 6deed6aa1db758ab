#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 4 / README from a bench.py JSON line.  Usage: bench_table.py BENCH.json"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
e = d["extra"]
c4, rb = e["c4"], e["c4"]["roofline_bwd"]
ceil = d["roofline"]["sustained_mfma_peak_measured"]["tflops"]
row = lambda name, ms, tf: f"| {name} | {ms:.2f} | {tf:.0f} | {tf / 25:.1f} | {tf / ceil * 100:.1f} |"
print(f"measured pure-MFMA rate in this run: {ceil:.0f} TFLOP/s")
print(row("C3 fwd b4 s16k causal fp16 (headline)", d["ms_per_step"], d["value"]))
print(row("C2 fwd b4 s4k fp16", e["c2"]["fwd_ms"], e["c2"]["fwd_tflops"]))
print(row("C5 shard fwd b4 s16k non-causal", e["c5shard"]["fwd_ms"], e["c5shard"]["fwd_tflops"]))
print(row("C4 fwd (bf16 s8k)", c4["fwd_ms"], c4["fwd_tflops"]))
print(row("C4 bwd (dQ incl. D + dK/dV)", c4["bwd_ms"], c4["bwd_tflops"]))
print(row("C4 fwd+bwd", c4["fwd_ms"] + c4["bwd_ms"], c4["fwd_bwd_tflops"]))
cb = d['cpu_baseline']
print(f"CPU baseline ({cb.get('kind')}): {cb['value']:.3f} TFLOP/s on {cb['cores']} threads" + (f"; C-oracle port {cb['oracle_port']['value']:.3f} TFLOP/s on {cb['oracle_port']['cores']} threads" if 'oracle_port' in cb else ""))
print("backward kernels alone:", {k: (round(v["avg_launch_ms"], 3), round(v["achieved"])) for k, v in rb.items()})
g = e["gqa_bwd_b4_s8192_d128_bf16_causal"]
print("GQA/MQA causal bwd ms:", {k: (round(v["bwd_ms"], 2), round(v["bwd_ms_without_workspace"], 2), round(v["vs_mha"], 2)) for k, v in g.items()})
for k in ("sweep_b4_h32_d128_fp16_noncausal", "sweep_b4_h32_d128_fp16_causal"):
    print(k, "TF", " / ".join(f"{v['tflops']:.0f}" for v in e[k].values()), "| vs SDPA", " / ".join(f"{v.get('speedup_vs_torch_sdpa', 0):.2f}" for v in e[k].values()))
