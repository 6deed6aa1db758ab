#!/usr/bin/env python3
"""Efficiency sweep across the configuration space, forward AND backward, through the C ABI: looks for OUTLIERS against
relations that should hold (causal ~0.5-0.55x the non-causal time at long s; varlen with equal lengths ~ the dense batch;
GQA ~ MHA; bf16 ~ fp16), because a defect in one kernel instantiation (round 1: the causal dK/dV accumulator shuffle) does
not show up in parity tests or in the one or two shapes a benchmark quotes.  Random data; median of `--rounds` timings."""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi  # noqa: E402
LIB = os.path.join(ROOT, "flash-attention-turing_amd", "csrc", "libflash_attn_gfx950.so")
vp, i32 = ctypes.c_void_p, ctypes.c_int


def load():
    L = ctypes.CDLL(LIB)
    L.fa_mha_fwd.argtypes = [vp] * 5 + [i32] * 8 + [vp]
    L.fa_mha_bwd.argtypes = [vp] * 10 + [i32] * 8 + [vp]
    L.fa_mha_varlen_fwd.argtypes = [vp] * 7 + [i32] * 8 + [vp]
    L.fa_mha_varlen_bwd.argtypes = [vp] * 12 + [i32] * 8 + [vp]
    for n in ("fa_mha_fwd", "fa_mha_bwd", "fa_mha_varlen_fwd", "fa_mha_varlen_bwd"):
        getattr(L, n).restype = i32
    L.fa_last_error.restype = ctypes.c_char_p
    return L


def timeit(fn, rounds, iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    L, dev = load(), torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    # name, b, sq, sk, h, hk, d, dtype, varlen
    F16, BF16 = torch.float16, torch.bfloat16
    shapes = [("8k d128 fp16", 4, 8192, 8192, 32, 32, 128, F16), ("8k d128 bf16", 4, 8192, 8192, 32, 32, 128, BF16),
              ("8k d64 fp16", 4, 8192, 8192, 32, 32, 64, F16), ("8k d64 bf16", 4, 8192, 8192, 32, 32, 64, BF16),
              ("2k d128 fp16", 16, 2048, 2048, 32, 32, 128, F16), ("512 d128 fp16", 64, 512, 512, 32, 32, 128, F16),
              ("512 d64 fp16", 64, 512, 512, 32, 32, 64, F16), ("128 d128 fp16", 256, 128, 128, 32, 32, 128, F16),
              ("8k GQA 32/8 d128 fp16", 4, 8192, 8192, 32, 8, 128, F16), ("8k MQA 32/1 d128 fp16", 4, 8192, 8192, 32, 1, 128, F16),
              ("8k GQA 32/8 d64 bf16", 4, 8192, 8192, 32, 8, 64, BF16),
              ("sq1k sk8k d128 fp16", 16, 1024, 8192, 32, 32, 128, F16), ("sq8k sk1k d128 fp16", 16, 8192, 1024, 32, 32, 128, F16),
              ("ragged 4000x4100 d128 fp16", 4, 4000, 4100, 32, 32, 128, F16), ("ragged 1000x1000 d64 bf16", 16, 1000, 1000, 32, 32, 64, BF16)]
    rows = {}
    print(f"{'shape':28s} {'causal':6s} {'mode':6s} | {'fwd ms':>8s} {'fwd TF':>7s} | {'bwd ms':>8s} {'bwd TF':>7s}", flush=True)
    for name, b, sq, sk, h, hk, d, dt in shapes:
        for causal in (False, True):
            for mode in ("dense", "varlen"):
                if mode == "varlen" and not (name.startswith("8k d128 fp16") or name.startswith("2k") or name.startswith("512 d128") or "GQA 32/8 d128" in name or name.startswith("ragged 1000")):
                    continue
                gen = torch.Generator(device=dev).manual_seed(1)
                mk = lambda n, s_, hh: torch.randn(n, s_, hh, d, device=dev, dtype=dt, generator=gen)
                q, k, v, do = mk(b, sq, h), mk(b, sk, hk), mk(b, sk, hk), mk(b, sq, h)
                o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
                lse = torch.empty(b, h, sq, device=dev, dtype=torch.float32)
                dsum = torch.empty(b, h, sq, device=dev, dtype=torch.float32)
                code = 0 if dt == F16 else 1
                if mode == "dense":
                    fwd = lambda: L.fa_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), b, sq, sk, h, hk, d, code, int(causal), st)
                    # parameter-struct entry point so that GQA / MQA shapes get the dK/dV workspace (head-group split, C ABI 3)
                    pb = capi.bwd_params(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
                    ws = capi.attach_workspace(pb, q)      # noqa: F841  (kept alive by the closure below)
                    bwd = lambda pb=pb, ws=ws: capi.lib().fa_run_mha_bwd(ctypes.byref(pb), st)
                else:   # same data seen as a packed batch of b equal-length sequences
                    cq = torch.arange(0, (b + 1) * sq, sq, device=dev, dtype=torch.int32)
                    ck = torch.arange(0, (b + 1) * sk, sk, device=dev, dtype=torch.int32)
                    fwd = lambda: L.fa_mha_varlen_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), cq.data_ptr(), ck.data_ptr(), b, sq, sk,
                                                      h, hk, d, code, int(causal), st)
                    bwd = lambda: L.fa_mha_varlen_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                                      dv.data_ptr(), dsum.data_ptr(), cq.data_ptr(), ck.data_ptr(), b, sq, sk, h, hk, d, code, int(causal), st)
                for f in (fwd, bwd):
                    rc = f()
                    assert rc == 0, (name, mode, causal, rc, L.fa_last_error())
                # visible (query, key) pairs: bottom-right aligned causal like the reference
                if causal:
                    dl = sk - sq
                    pairs = sum(max(0, min(sk, i + dl + 1)) for i in range(sq))
                else:
                    pairs = sq * sk
                fl = 4.0 * b * h * pairs * d
                iters = 3 if fl > 2e12 else 10
                tf, tb = timeit(fwd, a.rounds, iters), timeit(bwd, a.rounds, iters)
                rows[(name, causal, mode)] = (tf, tb)
                print(f"{name:28s} {str(causal):6s} {mode:6s} | {tf:8.3f} {fl / tf / 1e9:7.0f} | {tb:8.3f} {2.5 * fl / tb / 1e9:7.0f}", flush=True)
    print("\nrelations (time ratios; expected in brackets):", flush=True)
    for (name, causal, mode), (tf, tb) in rows.items():
        if causal and (name, False, mode) in rows and mode == "dense":
            nf, nb = rows[(name, False, mode)]
            print(f"  causal / non-causal  {name:28s} fwd {tf / nf:5.2f}  bwd {tb / nb:5.2f}   [~0.5-0.6 when sq==sk and long; ~1 when sk >> sq]", flush=True)
    for (name, causal, mode), (tf, tb) in rows.items():
        if mode == "varlen":
            df, db = rows[(name, causal, "dense")]
            print(f"  varlen / dense       {name:28s} causal={causal!s:5s} fwd {tf / df:5.2f}  bwd {tb / db:5.2f}   [~1.0]", flush=True)
    for a_, b_ in (("8k d128 bf16", "8k d128 fp16"), ("8k d64 bf16", "8k d64 fp16"), ("8k GQA 32/8 d128 fp16", "8k d128 fp16"), ("8k MQA 32/1 d128 fp16", "8k d128 fp16")):
        for causal in (False, True):
            (tf, tb), (nf, nb) = rows[(a_, causal, "dense")], rows[(b_, causal, "dense")]
            print(f"  {a_:22s} / {b_:14s} causal={causal!s:5s} fwd {tf / nf:5.2f}  bwd {tb / nb:5.2f}   [~1.0]", flush=True)


if __name__ == "__main__":
    main()
