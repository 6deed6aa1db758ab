#!/usr/bin/env python3
"""fwd + bwd of one varlen batch N times (for rocprofv3 --kernel-trace --stats): run_varlen_once.py "8192,64x63" causal [iters]
   (a dense call is "8192" with batch given as "8192x1")"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention-turing_amd"))
from flash_attn_turing import capi  # noqa: E402

lengths = []
for part in sys.argv[1].split(","):
    n, _, rep = part.partition("x")
    lengths += [int(n)] * (int(rep) if rep else 1)
causal = bool(int(sys.argv[2]))
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
h = hk = 32; d = 128; dt = torch.float16
DEV = torch.device("cuda:0")
L = capi.lib(); st = torch.cuda.current_stream(DEV).cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())
tot, b, mx = sum(lengths), len(lengths), max(lengths)
gen = torch.Generator(device=DEV).manual_seed(3)
q, do = (torch.randn(tot, h, d, device=DEV, dtype=dt, generator=gen) for _ in range(2))
k, v = (torch.randn(tot, hk, d, device=DEV, dtype=dt, generator=gen) for _ in range(2))
o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
lse = torch.zeros(b, h, mx, device=DEV, dtype=torch.float32); dsum = torch.zeros(b, h, mx, device=DEV, dtype=torch.float32)
cu = torch.tensor([0] + list(torch.tensor(lengths).cumsum(0)), device=DEV, dtype=torch.int32)
code = capi.dtype_code(dt)
for _ in range(iters):
    capi.check(L.fa_mha_varlen_fwd(p(q), p(k), p(v), p(o), p(lse), p(cu), p(cu), b, mx, mx, h, hk, d, code, int(causal), st))
    capi.check(L.fa_mha_varlen_bwd(p(q), p(k), p(v), p(o), p(lse), p(do), p(dq), p(dk), p(dv), p(dsum), p(cu), p(cu), b, mx, mx, h, hk, d, code, int(causal), st))
torch.cuda.synchronize()
print("done")
