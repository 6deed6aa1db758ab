#!/usr/bin/env python3
"""The two head_dim-128 kernel sets against each other, per stage, over a sequence ladder: one library, fa_set_kernel_policy switched between
interleaved timings (development aid; the thresholds of FA_POLICY_AUTO come from tables like this one).  Usage: ab_policy_sweep.py [--h 32] [--dtype fp16]"""
import argparse, ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=4); ap.add_argument("--h", type=int, default=32); ap.add_argument("--dtype", default="fp16")
ap.add_argument("--d", type=int, default=128); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--seqs", default="1024,2048,4096,8192,16384")
a = ap.parse_args()
dev = torch.device("cuda:0"); dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
worst = 1.0
for causal in (False, True):
    for s in (int(x) for x in a.seqs.split(",")):
        g = torch.Generator(device=dev).manual_seed(s)
        q, k, v, do = (torch.randn(a.b, s, a.h, a.d, device=dev, dtype=dt, generator=g) for _ in range(4))
        o, dq, dk, dv = (torch.empty_like(q) for _ in range(4)); lse, dsum = (torch.empty(a.b, a.h, s, device=dev, dtype=torch.float32) for _ in range(2))
        pb = capi.bwd_params(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
        capi.set_kernel_policy(capi.POLICY_AUTO)
        capi.mha_fwd(q, k, v, o, lse, causal); capi.bwd_stage("dq", pb); torch.cuda.synchronize()
        auto = {st: capi.kernel_name(st, a.b, s, s, a.h, a.d, causal, a.dtype) for st in ("fwd", "dq", "dkdv")}
        row = []
        for st in ("fwd", "dq", "dkdv"):
            f = (lambda: capi.mha_fwd(q, k, v, o, lse, causal)) if st == "fwd" else (lambda: capi.bwd_stage(st, pb))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); f(); f(); e1.record(); e1.synchronize()
            iters = max(3, min(200, int(10.0 / max(e0.elapsed_time(e1) / 3, 1e-3))))      # windows of ~10 ms
            t = {0: [], 1: []}
            for _ in range(a.rounds):
                for pol in (0, 1):
                    capi.set_kernel_policy(pol); f(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters): f()
                    e1.record(); e1.synchronize(); t[pol].append(e0.elapsed_time(e1) / iters)
            m32, m16 = statistics.median(t[0]), statistics.median(t[1])
            pick16 = "16" in auto[st]
            r_auto = (m16 if pick16 else m32) / min(m16, m32)      # ratio_auto_over_best_pinned: AUTO launches exactly the kernel of the set it picks
            worst = max(worst, r_auto)
            row.append(f"{st} 32x32 {m32:8.3f} ms 16x16 {m16:8.3f} ms ({m16 / m32:5.3f}) auto->{'16' if pick16 else '32'} auto/best {r_auto:5.3f}{'' if r_auto <= 1.03 else ' (!)'}")
        print(f"b{a.b} h{a.h} s{s:6d} {a.dtype} causal={int(causal)} | " + " | ".join(row), flush=True)
capi.set_kernel_policy(capi.POLICY_AUTO)
print(f"# b{a.b} h{a.h} d{a.d} {a.dtype}: worst ratio_auto_over_best_pinned {worst:5.3f}  ('(!)' = above 1.03)")
