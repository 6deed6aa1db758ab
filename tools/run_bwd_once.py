#!/usr/bin/env python3
"""Run the backward N times on one shape (for `rocprofv3 --kernel-trace --stats`): run_bwd_once.py b s h hk d dtype causal [iters]"""
import sys
import os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention-turing_amd"))
from flash_attn_turing import capi  # noqa: E402

b, s, h, hk, d = (int(x) for x in sys.argv[1:6])
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[6]]
causal = bool(int(sys.argv[7]))
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = "cuda:0"
gen = torch.Generator(device=dev).manual_seed(1)
mk = lambda hh: torch.randn(b, s, hh, d, device=dev, dtype=dt, generator=gen)
q, k, v, do = mk(h), mk(hk), mk(hk), mk(h)
o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
lse = torch.empty(b, h, s, device=dev, dtype=torch.float32)
dsum = torch.empty(b, h, s, device=dev, dtype=torch.float32)
capi.mha_fwd(q, k, v, o, lse, causal)
for _ in range(iters):
    capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
torch.cuda.synchronize()
print("done")
