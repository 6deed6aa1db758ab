#!/usr/bin/env python3
"""Interleaved in-process A/B of the forward (fa_mha_fwd) between two or more builds of the library: rounds x builds,
median / min / max per build, random data, outputs compared with the first build.  Usage: ab_fwd.py A.so B.so [...]"""
import argparse
import ctypes
import statistics
import torch

vp, i32 = ctypes.c_void_p, ctypes.c_int
F16, BF16 = torch.float16, torch.bfloat16
CONFIGS = {"d64 8k causal fp16": (4, 8192, 32, 32, 64, F16, True), "d64 8k causal bf16": (4, 8192, 32, 32, 64, BF16, True),
           "d64 2k causal fp16": (16, 2048, 32, 32, 64, F16, True), "d64 512 causal fp16": (64, 512, 32, 32, 64, F16, True),
           "d64 8k GQA32/8 causal bf16": (4, 8192, 32, 8, 64, BF16, True), "d64 16k causal fp16": (2, 16384, 32, 32, 64, F16, True),
           "c3: d128 16k causal fp16": (4, 16384, 32, 32, 128, F16, True), "c2: d128 4k fp16": (4, 4096, 32, 32, 128, F16, False),
           "d64 8k non-causal fp16 (control)": (4, 8192, 32, 32, 64, F16, False), "d128 8k causal fp16 (control)": (4, 8192, 32, 32, 128, F16, True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="", help="substring filter on config names")
    a = ap.parse_args()
    libs = {}
    for i, p in enumerate(a.libs):
        L = ctypes.CDLL(p)
        L.fa_mha_fwd.argtypes = [vp] * 5 + [i32] * 8 + [vp]
        L.fa_mha_fwd.restype = i32
        libs[f"{chr(65 + i)}:" + p.split("/")[-1]] = L
    dev = torch.device("cuda:0")
    for cname, (b, s, h, hk, d, dt, causal) in CONFIGS.items():
        if a.only and a.only not in cname:
            continue
        gen = torch.Generator(device=dev).manual_seed(1)
        q = torch.randn(b, s, h, d, device=dev, dtype=dt, generator=gen)
        k = torch.randn(b, s, hk, d, device=dev, dtype=dt, generator=gen)
        v = torch.randn(b, s, hk, d, device=dev, dtype=dt, generator=gen)
        outs = {n: (torch.empty_like(q), torch.empty(b, h, s, device=dev, dtype=torch.float32)) for n in libs}
        st = torch.cuda.current_stream(dev).cuda_stream
        flops = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)

        def run(n):
            o, lse = outs[n]
            rc = libs[n].fa_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), b, s, s, h, hk, d, 0 if dt == F16 else 1, int(causal), st)
            assert rc == 0, rc
        for n in libs:
            run(n)
        torch.cuda.synchronize()
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
        na = list(libs)[0]
        ma = statistics.median(times[na])
        for n, ts in times.items():
            med = statistics.median(ts)
            same = all(torch.equal(x, y) for x, y in zip(outs[na], outs[n]))
            print(f"{cname:34s} {n:30s} median {med:8.3f} ms (min {min(ts):8.3f} max {max(ts):8.3f}) {flops / med / 1e9:7.0f} TF  time vs A {med / ma:6.4f}  bit-identical to A: {same}", flush=True)


if __name__ == "__main__":
    main()
