#!/usr/bin/env python3
"""Fixed cost of a forward launch: time b4 h32 d128 fp16 non-causal at seq 256 .. 2048 where every CU holds at most a few workgroups,
back-to-back launches on one stream (HIP events around 50 launches), and fit  t = a + b * (key tiles per workgroup) * (workgroups per CU).
a = what a launch costs before any steady-state tile (launch gap + Q load + first K / V tiles + ping-pong offset + epilogue)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi  # noqa: E402

dev = torch.device("cuda:0")
rows = []
for causal in (False, True):
    for s in (256, 512, 768, 1024, 1536, 2048):
        g = torch.Generator(device=dev).manual_seed(0)
        q, k, v = (torch.randn(4, s, 32, 128, device=dev, dtype=torch.float16, generator=g) for _ in range(3))
        o = torch.empty_like(q)
        lse = torch.empty(4, 32, s, device=dev, dtype=torch.float32)
        f = lambda: capi.mha_fwd(q, k, v, o, lse, causal)
        for _ in range(10):
            f()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                f()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        us = float(np.median(ts))
        q_tiles = (s + 255) // 256
        wgs_per_cu = q_tiles * 128 / 256.0
        key_tiles = s / 64.0 * (0.5 + 2.0 / (s / 64.0) if causal else 1.0)     # causal: mean over the query tiles of one head, incl. the diagonal band
        rows.append((causal, s, us, wgs_per_cu, key_tiles))
        print(f"causal={causal!s:5s} seq {s:5d}: {us:7.2f} us per launch   workgroups/CU {wgs_per_cu:4.1f}   key tiles/workgroup {key_tiles:5.1f}", flush=True)
for causal in (False, True):
    r = [x for x in rows if x[0] == causal]
    x = np.array([max(1.0, w) * t for _, _, _, w, t in r])           # tile-iterations the busiest CU runs
    y = np.array([u for _, _, u, _, _ in r])
    b, a = np.polyfit(x, y, 1)
    print(f"causal={causal}: fit  t = {a:.1f} us + {b:.2f} us per tile-iteration  (steady state at 8k: ~1.7 us per tile-iteration)")

# launch-to-launch floor on an in-order stream: one-workgroup forward (b1 s64 h1: a single 64-key tile) and a 1-element torch kernel
def per_launch(f, n=200):
    for _ in range(20):
        f()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(ts))


q1, k1, v1 = (torch.randn(1, 64, 1, 128, device=dev, dtype=torch.float16) for _ in range(3))
o1, l1 = torch.empty_like(q1), torch.empty(1, 1, 64, device=dev, dtype=torch.float32)
x = torch.zeros(1, device=dev)
print(f"one-workgroup forward (1 key tile): {per_launch(lambda: capi.mha_fwd(q1, k1, v1, o1, l1, False)):.2f} us per launch;  1-element torch add_: {per_launch(lambda: x.add_(1.0)):.2f} us per launch")
