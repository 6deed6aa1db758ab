#!/usr/bin/env python3
"""Read the per-wave phase cycle counters a -DFA_KV_TIMING build of the dK/dV kernels leaves in the workspace (development aid).
Usage: phase_timing_dkdv.py LIB.so [names of the 6 phases ...]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
names = sys.argv[2:] or ["dma issue", "S/dP mfma", "valu", "dV/dK mfma", "vmcnt", "barrier+edge"]
for n in ("fa_run_mha_fwd",):
    getattr(L, n).argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
for n in ("fa_run_mha_bwd", "fa_bwd_dkdv"):
    getattr(L, n).argtypes = [ctypes.POINTER(capi.BwdParams), ctypes.c_void_p]
dev = torch.device("cuda:0")
for (b, s, h, d, causal, dt) in ((4, 8192, 32, 128, False, torch.bfloat16), (4, 8192, 32, 128, True, torch.bfloat16), (4, 2048, 32, 128, False, torch.float16)):
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v, do = (torch.randn(b, s, h, d, device=dev, dtype=dt, generator=g) for _ in range(4))
    o, dq, dk, dv = (torch.empty_like(q) for _ in range(4))
    lse, dsum = (torch.empty(b, h, s, device=dev, dtype=torch.float32) for _ in range(2))
    st = torch.cuda.current_stream().cuda_stream
    assert L.fa_run_mha_fwd(ctypes.byref(capi.fwd_params(q, k, v, o, lse, causal)), st) == 0
    pb = capi.bwd_params(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
    assert L.fa_run_mha_bwd(ctypes.byref(pb), st) == 0
    grid = (s // 128) * b * h
    ws = torch.zeros(grid * 8 * 8, device=dev, dtype=torch.float32)
    pb.workspace, pb.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    for _ in range(3):
        assert L.fa_bwd_dkdv(ctypes.byref(pb), st) == 0
    torch.cuda.synchronize()
    t = ws.view(grid, 8, 8).double().cpu()
    for grp, gname in ((slice(0, 4), "waves 0-3 (qh 0)"), (slice(4, 8), "waves 4-7 (qh 1)")):
        tt = t[:, grp]
        n = tt[..., 6]
        m = n > 0
        per = [(tt[..., i][m] / n[m]).mean().item() for i in range(6)]
        print(f"b{b} s{s} causal={causal} {gname}: cycles/tile  " + "  ".join(f"{nm} {x:7.1f}" for nm, x in zip(names, per)) + f"  total {sum(per):7.1f}  (MFMA issue alone: 2 x 512 per wave, 2048 per SIMD)")
