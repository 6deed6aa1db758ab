#!/usr/bin/env python3
"""Varlen with very unequal lengths vs the dense call doing the same real work, forward and backward, in one process:
  plain grid   = flat C functions fa_mha_varlen_* (no token totals -> grid = tiles(max_seqlen) x batch x heads; short sequences launch
                 workgroups that load cu_seqlens and exit)
  compact grid = param-struct entry points fa_run_mha_fwd / bwd with total_q / total_k set (what the torch module passes) -> grid = (ceil(total / BM) + batch) x heads"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi  # noqa: E402

DEV = torch.device("cuda:0")


def med(fn, rounds=7, iters=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)


def run(lengths, h, hk, d, dt, causal):
    L = capi.lib()
    st = torch.cuda.current_stream(DEV).cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    tot, b, mx = sum(lengths), len(lengths), max(lengths)
    gen = torch.Generator(device=DEV).manual_seed(3)
    q, do = (torch.randn(tot, h, d, device=DEV, dtype=dt, generator=gen) for _ in range(2))
    k, v = (torch.randn(tot, hk, d, device=DEV, dtype=dt, generator=gen) for _ in range(2))
    o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lse = torch.zeros(b, h, mx, device=DEV, dtype=torch.float32)
    dsum = torch.zeros(b, h, mx, device=DEV, dtype=torch.float32)
    cu = torch.tensor([0] + list(torch.tensor(lengths).cumsum(0)), device=DEV, dtype=torch.int32)
    code = capi.dtype_code(dt)
    f = lambda: capi.check(L.fa_mha_varlen_fwd(p(q), p(k), p(v), p(o), p(lse), p(cu), p(cu), b, mx, mx, h, hk, d, code, int(causal), st))
    g = lambda: capi.check(L.fa_mha_varlen_bwd(p(q), p(k), p(v), p(o), p(lse), p(do), p(dq), p(dk), p(dv), p(dsum), p(cu), p(cu), b, mx, mx, h, hk, d,
                                               code, int(causal), st))
    # compact grid: the SAME buffers through the param-struct entry points, with total_q / total_k set (what the torch module does)
    def structs(total):
        S = capi.Strides
        row = lambda t: S(0, t.stride(0), t.stride(1))
        fp = capi.FwdParams(q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=o2.data_ptr(), lse=lse2.data_ptr(), cu_seqlens_q=cu.data_ptr(),
                            cu_seqlens_k=cu.data_ptr(), b=b, seqlen_q=mx, seqlen_k=mx, h=h, h_k=hk, d=d, dtype=code, is_causal=int(causal),
                            q_stride=row(q), k_stride=row(k), v_stride=row(v), o_stride=row(o2), total_q=total, total_k=total)
        bp = capi.BwdParams(q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=o2.data_ptr(), dout=do.data_ptr(), lse=lse2.data_ptr(), dq=dq2.data_ptr(),
                            dk=dk2.data_ptr(), dv=dv2.data_ptr(), dsoftmax_sum=dsum.data_ptr(), cu_seqlens_q=cu.data_ptr(), cu_seqlens_k=cu.data_ptr(),
                            b=b, seqlen_q=mx, seqlen_k=mx, h=h, h_k=hk, d=d, dtype=code, is_causal=int(causal), q_stride=row(q), k_stride=row(k),
                            v_stride=row(v), o_stride=row(o2), do_stride=row(do), dq_stride=row(dq2), dk_stride=row(dk2), dv_stride=row(dv2),
                            total_q=total, total_k=total)
        return fp, bp
    o2, dq2, dk2, dv2 = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    lse2 = torch.zeros_like(lse)
    fp, bp = structs(tot)
    fc = lambda: capi.check(L.fa_run_mha_fwd(ctypes.byref(fp), st))
    gc = lambda: capi.check(L.fa_run_mha_bwd(ctypes.byref(bp), st))
    f(); g(); fc(); gc()
    torch.cuda.synchronize()
    same = all(torch.equal(a_, b_) for a_, b_ in ((o, o2), (dq, dq2), (dk, dk2), (dv, dv2), (lse, lse2)))
    return med(f), med(g), med(fc), med(gc), same


def dense(bb, s, h, hk, d, dt, causal):
    gen = torch.Generator(device=DEV).manual_seed(3)
    q, do = (torch.randn(bb, s, h, d, device=DEV, dtype=dt, generator=gen) for _ in range(2))
    k, v = (torch.randn(bb, s, hk, d, device=DEV, dtype=dt, generator=gen) for _ in range(2))
    o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lse = torch.empty(bb, h, s, device=DEV, dtype=torch.float32)
    dsum = torch.empty(bb, h, s, device=DEV, dtype=torch.float32)
    return med(lambda: capi.mha_fwd(q, k, v, o, lse, causal)), med(lambda: capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, causal))


def main():
    h, hk, d, dt = 32, 32, 128, torch.float16
    print("times in ms (fwd, bwd); plain and compact use the same preallocated buffers")
    for causal in (False, True):
        for name, lengths, eq in (("1 x 8192 + 15 x 64", [8192] + [64] * 15, (1, 8192)),
                                  ("1 x 8192 + 63 x 64", [8192] + [64] * 63, (1, 8192)),
                                  ("2 x 4096 + 30 x 128", [4096] * 2 + [128] * 30, (2, 4096)),
                                  ("4 x 2048 + 60 x 32", [2048] * 4 + [32] * 60, (4, 2048)),
                                  ("60 x 32 + 4 x 2048 (long ones last)", [32] * 60 + [2048] * 4, (4, 2048)),
                                  ("8 x 1024 .. 8 x 128 mixed", [1024, 128, 512, 256] * 8, None),
                                  ("16 x 512 (equal, control)", [512] * 16, (16, 512))):
            vf, vb, cf, cb, same = run(lengths, h, hk, d, dt, causal)
            if eq is None:
                print(f"{('causal ' if causal else 'non-causal ') + name:44s} plain {vf:7.3f} {vb:7.3f} | compact {cf:7.3f} {cb:7.3f} | bit-identical {same}", flush=True)
                continue
            df, db = dense(eq[0], eq[1], h, hk, d, dt, causal)
            print(f"{('causal ' if causal else 'non-causal ') + name:44s} plain {vf:7.3f} {vb:7.3f} | compact {cf:7.3f} {cb:7.3f} | dense b{eq[0]} s{eq[1]} {df:7.3f} {db:7.3f} | "
                  f"plain/dense {vf / df:4.2f} {vb / db:4.2f}  compact/dense {cf / df:4.2f} {cb / db:4.2f}  bit-identical {same}", flush=True)

if __name__ == "__main__":
    main()
