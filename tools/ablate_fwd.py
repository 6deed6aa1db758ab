#!/usr/bin/env python3
"""Timing-only ablations of the shipped forward kernel (fa_fwd_pp.hip, -DFA_ABL=n): which phase is the pole?

Builds (here, no GPU needed: `--build`) one extra library per ablation under tools/abl/, then on the GPU
times all of them INTERLEAVED in one process (rounds x variants, median/min/max per variant), random data.
Ablated variants compute WRONG results by construction; only their launch durations are used.
  bit0 (1): softmax phase without v_exp        bit1 (2): softmax phase without fma/exp/row-sum
  bit2 (4): matrix phase with half the LDS fragment reads (every K / V fragment used for two MFMAs)
"""
import argparse
import ctypes
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flash-attention-turing_amd")
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(ROOT, "tools", "abl")
VARIANTS = {"base": 0, "noexp": 1, "nosoftmax": 2, "halflds": 4, "halflds+noexp": 5, "halflds+nosoftmax": 6, "band_as_full": 8, "band_noskip": 16, "no_epilogue": 32, "no_q_load": 64, "no_epilogue+no_q_load": 96}


def build(only=None):
    sys.path.insert(0, PKG)
    import build as b
    os.makedirs(OUT, exist_ok=True)
    for name, bits in VARIANTS.items():
        if only and name not in only:
            continue
        objs = []
        for src in b.HIP_SOURCES:
            o = os.path.join(OUT, f"{name}_{src}.o")
            cmd = [b.hipcc_path()] + b.HIPCC_FLAGS + [f"-DFA_ABL={bits}", "-I", CSRC, "-I", b.INCLUDE, "-c", os.path.join(CSRC, src), "-o", o]
            subprocess.check_call(cmd)
            objs.append(o)
        subprocess.check_call([b.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libfa_{name}.so")] + objs)
        for o in objs:
            os.remove(o)
        print("built", name, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--variants", default="", help="comma list (default: all)")
    ap.add_argument("--causal-sweep", action="store_true", help="causal d128 fp16 b4 h32 at 1k..16k instead of the default configs")
    ap.add_argument("--short-sweep", action="store_true", help="d128 fp16 b4 h32 (and b64 at 512) at 512..4k, non-causal and causal")
    a = ap.parse_args()
    if a.build:
        return build(a.variants.split(",") if a.variants else None)
    import torch
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    libs = {}
    names = [n for n in VARIANTS if not a.variants or n in a.variants.split(",")]
    for name in names:
        L = ctypes.CDLL(os.path.join(OUT, f"libfa_{name}.so"))
        L.fa_mha_fwd.argtypes = [vp] * 5 + [i32] * 8 + [vp]
        L.fa_mha_fwd.restype = i32
        libs[name] = L
    dev = torch.device("cuda:0")
    cfgs = {"c3 fp16 d128 causal 16k": (4, 16384, 32, 128, torch.float16, True),
            "nc fp16 d128 8k": (4, 8192, 32, 128, torch.float16, False),
            "nc fp16 d64 8k": (4, 8192, 32, 64, torch.float16, False)}
    if a.causal_sweep:
        cfgs = {f"causal fp16 d128 {s_}": (4, s_, 32, 128, torch.float16, True) for s_ in (1024, 2048, 4096, 8192, 16384)}
    if a.short_sweep:
        cfgs = {}
        for s_ in (512, 1024, 2048, 4096):
            cfgs[f"nc fp16 d128 b4 {s_}"] = (4, s_, 32, 128, torch.float16, False)
            cfgs[f"causal fp16 d128 b4 {s_}"] = (4, s_, 32, 128, torch.float16, True)
        cfgs["nc fp16 d128 b64 512"] = (64, 512, 32, 128, torch.float16, False)
    for cname, (b, s, h, d, dt, causal) in cfgs.items():
        gen = torch.Generator(device=dev).manual_seed(1)
        q, k, v = (torch.randn(b, s, h, d, device=dev, dtype=dt, generator=gen) for _ in range(3))
        o = torch.empty_like(q)
        lse = torch.empty(b, h, s, device=dev, dtype=torch.float32)
        flops = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)
        st = torch.cuda.current_stream(dev).cuda_stream

        def run(L):
            rc = L.fa_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), b, s, s, h, h, d, 0, int(causal), st)
            assert rc == 0, rc
        for L in libs.values():
            run(L)
        torch.cuda.synchronize()
        times = {n: [] for n in libs}
        for _ in range(a.rounds):
            for n, L in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(L)
                e1.record()
                e1.synchronize()
                times[n].append(e0.elapsed_time(e1) / a.iters)
        base = statistics.median(times["base"])
        for n, ts in times.items():
            med = statistics.median(ts)
            print(f"{cname:26s} {n:20s} median {med:8.3f} ms (min {min(ts):8.3f} max {max(ts):8.3f})  {flops / med / 1e9:7.1f} 'TF'  time vs base {med / base:6.3f}", flush=True)


if __name__ == "__main__":
    main()
