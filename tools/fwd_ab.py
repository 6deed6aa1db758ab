#!/usr/bin/env python3
"""In-process interleaved A/B timing of the forward schedules (cdna guide rule 24):
N variants x M rounds in ONE process, median / min per variant, random data."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
import torch  # noqa: E402
from flash_attn_turing import capi  # noqa: E402

CONFIGS = {
    "c2": (4, 4096, 32, 32, 128, torch.float16, False),
    "c3": (4, 16384, 32, 32, 128, torch.float16, True),
    "nc16k": (4, 16384, 32, 32, 128, torch.float16, False),
    "c8k": (4, 8192, 32, 32, 128, torch.float16, True),
    "bf16_8k": (4, 8192, 32, 32, 128, torch.bfloat16, False),
    "d64_8k": (4, 8192, 32, 32, 64, torch.float16, False),
    "gqa_8k": (4, 8192, 32, 8, 128, torch.float16, True),
    "s1k": (4, 1024, 32, 32, 128, torch.float16, False),
}
# the north_star's sweep: b4 h32 d128 fp16, seq 512..16k, non-causal and causal (+ d64 and a larger batch at 512)
for _s in (512, 1024, 2048, 4096, 8192, 16384):
    CONFIGS[f"nc{_s}"] = (4, _s, 32, 32, 128, torch.float16, False)
    CONFIGS[f"ca{_s}"] = (4, _s, 32, 32, 128, torch.float16, True)
    CONFIGS[f"d64nc{_s}"] = (4, _s, 32, 32, 64, torch.float16, False)
    CONFIGS[f"d64ca{_s}"] = (4, _s, 32, 32, 64, torch.float16, True)
CONFIGS["b64nc512"] = (64, 512, 32, 32, 128, torch.float16, False)
CONFIGS["b64ca512"] = (64, 512, 32, 32, 128, torch.float16, True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c2,c3,nc16k")
    ap.add_argument("--impls", default="simple,pp")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in a.configs.split(","):
        b, s, h, hk, d, dt, causal = CONFIGS[name]
        gen = torch.Generator(device=dev).manual_seed(1)
        q = torch.randn(b, s, h, d, device=dev, dtype=dt, generator=gen)
        k = torch.randn(b, s, hk, d, device=dev, dtype=dt, generator=gen)
        v = torch.randn(b, s, hk, d, device=dev, dtype=dt, generator=gen)
        o = torch.empty_like(q)
        lse = torch.empty(b, h, s, device=dev, dtype=torch.float32)
        flops = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)
        times = {i: [] for i in a.impls.split(",")}
        for impl in times:
            capi.set_fwd_impl(impl)
            capi.mha_fwd(q, k, v, o, lse, causal)
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for impl in times:
                capi.set_fwd_impl(impl)
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(a.iters):
                    capi.mha_fwd(q, k, v, o, lse, causal)
                en.record()
                en.synchronize()
                times[impl].append(st.elapsed_time(en) / a.iters)
        for impl, ts in times.items():
            med, mn = statistics.median(ts), min(ts)
            print(f"{name:8s} {impl:7s} median {med:8.3f} ms  {flops / med / 1e9:7.1f} TF   min {mn:8.3f} ms  {flops / mn / 1e9:7.1f} TF", flush=True)
    capi.set_fwd_impl(None)


if __name__ == "__main__":
    main()
