#!/usr/bin/env python3
"""Interleaved in-process A/B of the whole backward (dot + dQ + dK/dV through fa_mha_bwd) between two builds of the
library: rounds x {A, B}, median / min / max per build, random data.  Usage: ab_bwd.py A.so B.so [C.so ...] [--rounds N]"""
import argparse
import ctypes
import statistics
import torch

vp, i32 = ctypes.c_void_p, ctypes.c_int


def load(path):
    L = ctypes.CDLL(path)
    L.fa_mha_fwd.argtypes = [vp] * 5 + [i32] * 8 + [vp]
    L.fa_mha_bwd.argtypes = [vp] * 10 + [i32] * 8 + [vp]
    L.fa_mha_fwd.restype = L.fa_mha_bwd.restype = i32
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+", help="two or more builds of libflash_attn_gfx950.so; the first is the baseline")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="", help="substring filter on config names")
    a = ap.parse_args()
    libs = {f"{chr(65 + i)}:" + p.split("/")[-1]: load(p) for i, p in enumerate(a.libs)}
    dev = torch.device("cuda:0")
    cfgs = {"c4 bf16 d128 8k": (4, 8192, 32, 32, 128, torch.bfloat16, False),
            "bf16 d128 8k causal": (4, 8192, 32, 32, 128, torch.bfloat16, True),
            "fp16 d64 8k": (4, 8192, 32, 32, 64, torch.float16, False),
            "fp16 d128 gqa 4k causal": (4, 4096, 32, 8, 128, torch.float16, True),
            "bf16 d128 1k causal": (16, 1024, 32, 32, 128, torch.bfloat16, True),
            "bf16 d128 2k causal": (8, 2048, 32, 32, 128, torch.bfloat16, True),
            "fp16 d128 16k causal (c3 bwd)": (4, 16384, 32, 32, 128, torch.float16, True),
            "fp16 d64 8k causal": (4, 8192, 32, 32, 64, torch.float16, True),
            "bf16 d64 2k": (16, 2048, 32, 32, 64, torch.bfloat16, False),
            "bf16 d64 2k causal": (16, 2048, 32, 32, 64, torch.bfloat16, True),
            "fp16 d64 512 causal": (64, 512, 32, 32, 64, torch.float16, True),
            "bf16 d64 8k GQA32/8 causal": (4, 8192, 32, 8, 64, torch.bfloat16, True),
            "fp16 d128 ragged 4000x4100": (4, 4000, 32, 32, 128, torch.float16, False)}
    for cname, (b, s, h, hk, d, dt, causal) in cfgs.items():
        if a.only and a.only not in cname and "c4" not in cname:
            continue
        sk = 4100 if "ragged" in cname else s
        gen = torch.Generator(device=dev).manual_seed(1)
        q = torch.randn(b, s, h, d, device=dev, dtype=dt, generator=gen)
        k = torch.randn(b, sk, hk, d, device=dev, dtype=dt, generator=gen)
        v = torch.randn(b, sk, hk, d, device=dev, dtype=dt, generator=gen)
        do = torch.randn(b, s, h, d, device=dev, dtype=dt, generator=gen)
        o = torch.empty_like(q)
        lse = torch.empty(b, h, s, device=dev, dtype=torch.float32)
        dsum = torch.empty(b, h, s, device=dev, dtype=torch.float32)
        st = torch.cuda.current_stream(dev).cuda_stream
        code = 0 if dt == torch.float16 else 1
        first = next(iter(libs.values()))
        assert first.fa_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), b, s, sk, h, hk, d, code, int(causal), st) == 0
        outs = {}
        for n, L in libs.items():
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            outs[n] = (dq, dk, dv)

        def run(n):
            dq, dk, dv = outs[n]
            rc = libs[n].fa_mha_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(),
                                    dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dsum.data_ptr(), b, s, sk, h, hk, d, code, int(causal), st)
            assert rc == 0, rc
        for n in libs:
            run(n)
        torch.cuda.synchronize()
        na = list(libs)[0]
        times = {n: [] for n in libs}
        for _ in range(a.rounds):
            for n in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(n)
                e1.record()
                e1.synchronize()
                times[n].append(e0.elapsed_time(e1) / a.iters)
        ma = statistics.median(times[na])
        for n in libs:
            ts = times[n]
            same = all(torch.equal(x, y) for x, y in zip(outs[na], outs[n]))
            md = max(float((x.float() - y.float()).abs().max()) for x, y in zip(outs[na], outs[n]))
            print(f"{cname:30s} {n:30s} median {statistics.median(ts):8.3f} ms (min {min(ts):8.3f} max {max(ts):8.3f})  time vs A {statistics.median(ts) / ma:6.4f}  "
                  f"bit-identical to A: {same} (max abs diff {md:.3g})", flush=True)


if __name__ == "__main__":
    main()
