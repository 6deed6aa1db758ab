#!/usr/bin/env python3
"""Forward-only interleaved A/B of library builds over a sequence-length ladder (development aid).
Usage: ab_fwd_seqs.py A.so B.so [--d 64] [--b 4] [--h 32] [--dtype fp16]"""
import argparse, ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+"); ap.add_argument("--d", type=int, default=64); ap.add_argument("--b", type=int, default=4)
ap.add_argument("--h", type=int, default=32); ap.add_argument("--dtype", default="fp16"); ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
libs = {}
for i, p in enumerate(a.libs):
    L = ctypes.CDLL(os.path.abspath(p)); L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
    libs[f"{chr(65 + i)}:{os.path.basename(p)[6:-3]}"] = L
dev = torch.device("cuda:0"); dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
for causal in (False, True):
    for s in (256, 512, 1024, 2048, 4096, 8192, 16384):
        g = torch.Generator(device=dev).manual_seed(s)
        q, k, v = (torch.randn(a.b, s, a.h, a.d, device=dev, dtype=dt, generator=g) for _ in range(3))
        o = torch.empty_like(q); lse = torch.empty(a.b, a.h, s, device=dev, dtype=torch.float32)
        pf = capi.fwd_params(q, k, v, o, lse, causal); st = torch.cuda.current_stream().cuda_stream
        iters = 20 if s <= 2048 else 5
        times = {n: [] for n in libs}
        for n, L in libs.items():
            assert L.fa_run_mha_fwd(ctypes.byref(pf), st) == 0
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for n, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): L.fa_run_mha_fwd(ctypes.byref(pf), st)
                e1.record(); e1.synchronize(); times[n].append(e0.elapsed_time(e1) / iters)
        base = statistics.median(times[list(libs)[0]])
        fl = 4.0 * a.b * a.h * s * s * a.d * (0.5 if causal else 1.0)
        print(f"d{a.d} b{a.b} h{a.h} s{s:6d} causal={int(causal)}  " + "  ".join(f"{n} {statistics.median(t) * 1e3:8.1f} us {fl / statistics.median(t) / 1e9:6.0f} TF x{statistics.median(t) / base:5.3f}" for n, t in times.items()), flush=True)
