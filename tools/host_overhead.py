#!/usr/bin/env python3
"""Host time per call of every entry level (C ABI through ctypes, torch binding, autograd wrappers) against the GPU time of the same calls, at the short shapes
where the host can be the bound; PyTorch's own SDPA + backward beside it.  Usage: host_overhead.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention-turing_amd"))
import flash_attn_turing as F
from flash_attn_turing import capi
dev = torch.device("cuda:0")
def t_host(f, n=300):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for (b, s, h, d) in ((4, 512, 32, 128), (1, 512, 8, 128)):
    q, k, v, do = (torch.randn(b, s, h, d, device=dev, dtype=torch.float16) for _ in range(4))
    o, lse = F.fwd(q, k, v, True)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3)); dsum = torch.empty_like(lse)
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
    def fb_acc():
        F.flash_attn_func(qg, kg, vg, causal=True).backward(do)
    def fb_none():
        qg.grad = kg.grad = vg.grad = None
        F.flash_attn_func(qg, kg, vg, causal=True).backward(do)
    def fb_grad():
        out = F.flash_attn_func(qg, kg, vg, causal=True)
        torch.autograd.grad(out, (qg, kg, vg), do)
    def sdpa():
        qg.grad = kg.grad = vg.grad = None
        torch.nn.functional.scaled_dot_product_attention(qg.transpose(1, 2), kg.transpose(1, 2), vg.transpose(1, 2), is_causal=True).backward(do.transpose(1, 2))
    rows = [("capi.mha_bwd (preallocated)", lambda: capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, True)), ("F.bwd (torch binding)", lambda: F.bwd(q, k, v, o, lse, do, True)),
            ("F.fwd + F.bwd", lambda: F.bwd(q, k, v, *F.fwd(q, k, v, True), do, True)),
            ("flash_attn_func + backward, .grad accumulating", fb_acc), ("flash_attn_func + backward, .grad = None first", fb_none), ("flash_attn_func + autograd.grad", fb_grad),
            ("torch SDPA + backward, .grad = None first", sdpa)]
    for name, f in rows:
        host, total = t_host(f)
        print(f"b{b} s{s} h{h} d{d} {name:48s} host issue {host:7.1f} us/call, wall {total:7.1f} us/call", flush=True)
