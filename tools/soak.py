#!/usr/bin/env python3
"""Determinism soak: forward + backward repeated N times per shape on one GPU, every result compared BITWISE with the first.
The kernels are atomics-free and every reduction has a fixed order, so any difference is a race (LDS ring reuse, hand-counted
vmcnt / barrier pairs, hidden LDS-DMA, the wave-local epilogue, the dK/dV workspace planes).  Usage: soak.py [--iters 300]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
import flash_attn_turing as F  # noqa: E402

SHAPES = [  # b, sq, sk, h, hk, d, dtype, causal
    (2, 4096, 4096, 16, 16, 128, torch.float16, True),
    (2, 4096, 4096, 16, 16, 128, torch.bfloat16, False),
    (1, 8192, 8192, 32, 1, 128, torch.bfloat16, True),      # MQA: dK/dV head-group split through the workspace
    (3, 1000, 1300, 8, 2, 128, torch.float16, True),        # ragged, GQA, sk > sq
    (3, 1300, 700, 6, 3, 64, torch.bfloat16, True),         # sq > sk: dead rows
    (4, 2048, 2048, 32, 32, 64, torch.float16, False),
    (64, 512, 512, 8, 8, 128, torch.float16, True),         # many short workgroups
    (1, 16384, 16384, 8, 8, 128, torch.float16, True),      # the headline's per-head problem (the 16x16x32 forward under every policy but mfma32)
    (2, 300, 4100, 4, 2, 128, torch.bfloat16, False),       # few rows, long ragged key axis
    (2, 4096, 4096, 16, 16, 64, torch.float16, False),      # head_dim 64 on the 16x16x32 forward under the default policy (round 4)
    (1, 8192, 8300, 8, 4, 64, torch.float16, True),         # the same under a mask, ragged key axis
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--policy", default="auto", choices=["auto", "mfma16", "mfma32"], help="head_dim-128 kernel set (fa_set_kernel_policy)")
    a = ap.parse_args()
    from flash_attn_turing import capi
    capi.set_kernel_policy({"auto": capi.POLICY_AUTO, "mfma16": capi.POLICY_MFMA16, "mfma32": capi.POLICY_MFMA32}[a.policy])
    print(f"policy {a.policy}: d128 kernels at 4096 non-causal:", [capi.kernel_name(st, 2, 4096, 4096, 16, 128, False) for st in ("fwd", "dq", "dkdv")], flush=True)
    dev = torch.device("cuda:0")
    bad = 0
    for (b, sq, sk, h, hk, d, dt, causal) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(5)
        q, do = (torch.randn(b, sq, h, d, device=dev, dtype=dt, generator=g) for _ in range(2))
        k, v = (torch.randn(b, sk, hk, d, device=dev, dtype=dt, generator=g) for _ in range(2))
        ref = None
        diffs = 0
        for i in range(a.iters):
            o, lse = F.fwd(q, k, v, causal)
            dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
            cur = (o, lse, dq, dk, dv)
            if ref is None:
                ref = tuple(t.clone() for t in cur)
            elif not all(torch.equal(x, y) for x, y in zip(cur, ref)):
                diffs += 1
        torch.cuda.synchronize()
        bad += diffs
        print(f"b{b} sq{sq} sk{sk} h{h}/{hk} d{d} {str(dt)[6:]} causal={causal}: {a.iters} iterations, {diffs} differing from the first", flush=True)
    print("SOAK", "FAILED" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
