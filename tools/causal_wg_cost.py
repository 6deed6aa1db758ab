#!/usr/bin/env python3
"""What a CAUSAL forward workgroup costs when scheduling is out of the picture: seqlen_q = 256 (one query tile per (batch, head)), seqlen_k = S,
bottom-right aligned mask -> every workgroup runs S/64 - 4 steady tiles + 4 diagonal ones and all workgroups are equal; against the same launch
without the mask.  If the two agree per tile, what a full causal launch loses against the non-causal rate is scheduling / locality, not the kernel.
Usage: causal_wg_cost.py"""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
dev = torch.device("cuda:0")
def t_ms(f, iters=10, rounds=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for b, h in ((32, 32), (8, 32)):
    for S in (1024, 2048, 4096, 8192, 16384):
        g = torch.Generator(device=dev).manual_seed(S)
        q = torch.randn(b, 256, h, 128, device=dev, dtype=torch.float16, generator=g)
        k, v = (torch.randn(b, S, h, 128, device=dev, dtype=torch.float16, generator=g) for _ in range(2))
        o = torch.empty_like(q); lse = torch.empty(b, h, 256, device=dev, dtype=torch.float32)
        row = []
        for causal in (False, True):
            ms = t_ms(lambda: capi.mha_fwd(q, k, v, o, lse, causal))
            fl = 4.0 * b * h * 256 * (S - (128 if causal else 0)) * 128
            row.append(f"{'causal' if causal else 'dense '} {ms:8.3f} ms {fl / ms / 1e9:6.0f} TF {capi.kernel_name('fwd', b, 256, S, h, 128, causal)[7:11]}")
        print(f"b{b} h{h} sq256 sk{S:6d} ({b * h // 256} workgroups per CU) | " + " | ".join(row), flush=True)
        del q, k, v, o, lse
