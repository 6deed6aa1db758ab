#!/usr/bin/env python3
"""The reference's headline chart on MI355X: forward time of this library vs PyTorch SDPA on the SAME GPU
(reference README.md:7-18 / utils/Speed_Up.png: b=4, h=32, d=128, no mask, seq 512..16384, "around 2x
faster than PyTorch attention" on a T4; sweep definition benchmark.sh:17-23).  Prints a table and one JSON
line; PyTorch-ROCm's SDPA picks its own fused backend (flash / mem-efficient via AOTriton or CK)."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
import torch  # noqa: E402
import torch.nn.functional as TF  # noqa: E402
from flash_attn_turing import capi  # noqa: E402


def time_ms(fn, iters):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(iters):
            fn()
        en.record(); en.synchronize()
        ts.append(st.elapsed_time(en) / iters)
    return statistics.median(ts)


def main():
    dev = torch.device("cuda:0")
    rows = []
    for d in (128, 64):
        for causal in (False, True):
            for s in (512, 1024, 2048, 4096, 8192, 16384):
                b, h = 4, 32
                gen = torch.Generator(device=dev).manual_seed(s + d)
                q, k, v = (torch.randn(b, s, h, d, device=dev, dtype=torch.float16, generator=gen) for _ in range(3))
                o = torch.empty_like(q)
                lse = torch.empty(b, h, s, device=dev, dtype=torch.float32)
                ours = time_ms(lambda: capi.mha_fwd(q, k, v, o, lse, causal), 20 if s <= 4096 else 5)
                qt, kt, vt = (t.permute(0, 2, 1, 3) for t in (q, k, v))          # SDPA wants (b, h, s, d); strided view, no copy
                try:
                    sdpa = time_ms(lambda: TF.scaled_dot_product_attention(qt, kt, vt, is_causal=causal), 20 if s <= 4096 else 5)
                    ref = TF.scaled_dot_product_attention(qt, kt, vt, is_causal=causal).permute(0, 2, 1, 3)
                    err = (ref.float() - o.float()).abs().max().item()
                except Exception as e:  # noqa: BLE001
                    sdpa, err = float("nan"), float("nan")
                    print("SDPA failed:", str(e)[:120])
                flops = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)
                rows.append(dict(d=d, causal=causal, seq=s, ours_ms=ours, sdpa_ms=sdpa, speedup=sdpa / ours,
                                 ours_tflops=flops / ours / 1e9, sdpa_tflops=flops / sdpa / 1e9, max_abs_diff=err))
                r = rows[-1]
                print(f"d={d:3d} causal={int(causal)} seq={s:6d}  ours {ours:8.3f} ms {r['ours_tflops']:7.1f} TF | torch SDPA {sdpa:8.3f} ms "
                      f"{r['sdpa_tflops']:7.1f} TF | speed-up {r['speedup']:.2f}x | max|diff| {err:.1e}", flush=True)
    print(json.dumps({"sweep": "b4 h32 fp16 fwd, ours vs torch SDPA on the same MI355X", "torch": torch.__version__, "rows": rows}))


if __name__ == "__main__":
    main()
