#!/usr/bin/env python3
"""What HIP-graph replay buys on launch-bound shapes: forward + backward (4 kernel launches) through the C ABI, eager vs captured graph."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention-turing_amd"))
from flash_attn_turing import capi  # noqa: E402

DEV = torch.device("cuda:0")


def med(fn, rounds=9, iters=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


for b, s, h, d, causal in ((1, 128, 8, 64, True), (2, 256, 16, 128, True), (4, 512, 32, 128, True), (4, 1024, 32, 128, True), (4, 4096, 32, 128, True)):
    dt = torch.float16
    gen = torch.Generator(device=DEV).manual_seed(1)
    q, k, v, do = (torch.randn(b, s, h, d, device=DEV, dtype=dt, generator=gen) for _ in range(4))
    o, dq, dk, dv = (torch.empty_like(q) for _ in range(4))
    lse = torch.empty(b, h, s, device=DEV, dtype=torch.float32)
    dsum = torch.empty_like(lse)

    def step(stream=None):
        capi.mha_fwd(q, k, v, o, lse, causal, stream=stream)
        capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, causal, stream=stream)
    step(); torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            step(side.cuda_stream)
    torch.cuda.synchronize()
    te, tg = med(step), med(g.replay)
    print(f"b{b} s{s} h{h} d{d} causal fwd+bwd: eager {te:8.1f} us   graph replay {tg:8.1f} us   ratio {tg / te:5.2f}", flush=True)
