#!/usr/bin/env python3
"""Forward outputs of two library builds on a ladder of small / ragged / causal shapes (development aid): max |dO|, max |dLSE|, the
first rows that differ by more than a rounding or two.  Usage: ab_check.py A.so B.so [--dtype bf16]"""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
import ab_stage as A
from flash_attn_turing import capi
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs=2); ap.add_argument("--dtype", default="fp16"); a = ap.parse_args()
la, lb = A.load(a.libs[0]), A.load(a.libs[1])
dev = torch.device("cuda:0"); dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
tol = 2e-3 if a.dtype == "fp16" else 1.6e-2
for (b, sq, sk, h, d, causal, mag) in [(1, 256, 256, 1, 128, True, 1), (1, 512, 512, 2, 128, True, 1), (2, 300, 389, 4, 128, True, 1), (1, 389, 300, 2, 128, True, 1),
                                       (4, 1024, 1024, 32, 128, True, 1), (1, 1024, 1024, 2, 128, False, 1), (2, 300, 389, 4, 128, False, 1), (1, 2048, 2048, 2, 128, False, 8),
                                       (1, 2048, 2048, 2, 128, True, 30), (1, 64, 4096, 2, 128, False, 1), (1, 1, 1, 1, 128, False, 1), (1, 3, 700, 1, 128, True, 1)]:
    gen = torch.Generator(device=dev).manual_seed(1)
    q = (torch.randn(b, sq, h, d, device=dev, dtype=torch.float32, generator=gen) * mag).to(dt)
    k, v = (torch.randn(b, sk, h, d, device=dev, dtype=dt, generator=gen) for _ in range(2))
    res = []
    for L in (la, lb):
        o = torch.full_like(q, 7.0); lse = torch.full((b, h, sq), 7.0, device=dev, dtype=torch.float32)
        p = capi.fwd_params(q, k, v, o, lse, causal)
        assert L.fa_run_mha_fwd(ctypes.byref(p), torch.cuda.current_stream(dev).cuda_stream) == 0
        torch.cuda.synchronize(); res.append((o.float(), lse))
    do = (res[0][0] - res[1][0]).abs(); dl = (res[0][1] - res[1][1]).abs()
    bad = torch.nonzero(~(do.amax(-1) < tol))          # (b, row, h)
    print(f"b{b} sq{sq} sk{sk} h{h} causal={causal} |q|x{mag}: max|dO| {do.max().item():.3e} max|dLSE| {dl.max().item():.3e} rows off by > {tol}: {bad.shape[0]}", bad[:4].tolist())
    if bad.shape[0]:
        bi, r, hi = bad[0].tolist()
        print("   first such row: A", res[0][0][bi, r, hi, :4].tolist(), "B", res[1][0][bi, r, hi, :4].tolist(), "lse", res[0][1][bi, hi, r].item(), res[1][1][bi, hi, r].item())
