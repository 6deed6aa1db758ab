#!/usr/bin/env python3
"""Run the forward kernel a few times for profiling (rocprofv3 --kernel-trace / --pmc)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
import torch  # noqa: E402
from flash_attn_turing import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seq", type=int, default=8192)
ap.add_argument("--b", type=int, default=4)
ap.add_argument("--h", type=int, default=32)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--causal", type=int, default=0)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--bwd", type=int, default=0)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--lib", default=None, help="an A/B build from tools/build_variant.py instead of the in-tree library")
a = ap.parse_args()
if a.lib:
    capi.LIBRARY_PATH = os.path.abspath(a.lib)
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
gen = torch.Generator(device=dev).manual_seed(1)
q, k, v, do = (torch.randn(a.b, a.seq, a.h, a.d, device=dev, dtype=dt, generator=gen) for _ in range(4))
o = torch.empty_like(q)
lse = torch.empty(a.b, a.h, a.seq, device=dev, dtype=torch.float32)
dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(lse)
for _ in range(a.iters):
    capi.mha_fwd(q, k, v, o, lse, bool(a.causal))
    if a.bwd:
        capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, bool(a.causal))
torch.cuda.synchronize()
print("done")
