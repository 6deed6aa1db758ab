#!/usr/bin/env python3
"""Read the per-wave phase cycle counters a -DFA_KV_TIMING build of the dK/dV kernels leaves in the workspace (development aid).
Usage: phase_timing_dkdv.py LIB.so [names of the 6 phases ...]
       phase_timing_dkdv.py LIB.so --layout 16 [--mqa]     (fa_bwd_dkdv16.hip, round 6: 7 phases + prologue / epilogue cycles + wall-clock start / end per workgroup;
                                                             prints the per-phase table and, for causal shapes, what a workgroup costs against its tile count)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
LAYOUT16 = "--layout" in sys.argv
if LAYOUT16:
    names = ["top: dma(w0-3)+stat load", "S/dP mfma", "dma(w4-7)+next desc", "exp/mask/dS/cvt", "dV/dK mfma", "stat store+vmcnt(0)", "toggle+barrier"]
else:
    names = sys.argv[2:] or ["dma issue", "S/dP mfma", "valu", "dV/dK mfma", "vmcnt", "barrier+edge"]
NP, W = (7, 16) if LAYOUT16 else (6, 8)
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
    ws = torch.zeros(grid * 8 * W, device=dev, dtype=torch.float32)
    pb.workspace, pb.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    for _ in range(3):
        assert L.fa_bwd_dkdv(ctypes.byref(pb), st) == 0
    torch.cuda.synchronize()
    t = ws.view(grid, 8, W).double().cpu()
    for grp, gname in ((slice(0, 4), "waves 0-3 (qh 0)"), (slice(4, 8), "waves 4-7 (qh 1)")):
        tt = t[:, grp]
        n = tt[..., NP]
        m = n > 0
        per = [(tt[..., i][m] / n[m]).mean().item() for i in range(NP)]
        print(f"b{b} s{s} causal={causal} {gname}: cycles/tile  " + "  ".join(f"{nm} {x:7.1f}" for nm, x in zip(names, per)) + f"  total {sum(per):7.1f}  (MFMA issue alone: 2 x 512 per wave, 2048 per SIMD)")

    if LAYOUT16:
        # per workgroup (wave 0's stamps): tiles, loop cycles, prologue, epilogue, wall-clock span; fit  loop = a + b * tiles  and report the fixed cost in tile-times
        w0 = t[:, 0]
        n = w0[:, 7]; m = n > 0
        loop = w0[:, :7].sum(dim=1)
        pro, epi = w0[:, 8], w0[:, 9]
        span = (w0[:, 11] - w0[:, 10]) % float(1 << 24)                  # 100 MHz ticks
        import numpy as np
        nn, ll = n[m].numpy(), loop[m].numpy()
        if len(set(nn.tolist())) > 1:
            bfit, afit = np.polyfit(nn, ll, 1)
        else:
            bfit, afit = (ll / nn).mean(), 0.0
        print(f"  per workgroup: loop cycles = {afit:8.0f} + {bfit:7.1f} x tiles;  prologue {pro[m].mean().item():7.0f} cyc, epilogue {epi[m].mean().item():7.0f} cyc "
              f"(= {(pro[m].mean().item() + epi[m].mean().item() + afit) / bfit:5.2f} tile-times of fixed cost per workgroup; mean tiles per workgroup {nn.mean():6.1f})")
        # the launch on the wall clock: first start to last end, and the sum of workgroup spans per XCD / per CU slot
        st, en = w0[:, 10][m].numpy(), w0[:, 11][m].numpy()
        base = st.min(); rel_end = ((en - base) % float(1 << 24)); rel_st = ((st - base) % float(1 << 24))
        total = rel_end.max()
        busy = span[m].numpy().sum()
        print(f"  launch: {total / 100.0:8.1f} us first start -> last end; sum of workgroup spans {busy / 100.0 / 256:8.1f} us per CU (256 CUs) -> {busy / 256 / total:5.3f} of the launch a CU is occupied; "
              f"last workgroup starts at {rel_st.max() / total:5.3f}, 95 % of the work is done at {np.sort(rel_end)[int(0.95 * len(rel_end))] / total:5.3f}")
        if causal:
            tl = w0[:, 12][m].numpy()
            for lo, hi in ((0, 8), (8, 24), (24, 48), (48, 64)):
                sel = (tl >= lo) & (tl < hi)
                if sel.any():
                    print(f"    key blocks {lo:2d}-{hi - 1:2d}: mean tiles {nn[sel].mean():6.1f}  cycles per tile {(ll[sel] / nn[sel]).mean():7.1f}  span per tile {(span[m].numpy()[sel] / nn[sel]).mean() * 10:7.1f} ns")
