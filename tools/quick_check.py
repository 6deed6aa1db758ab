#!/usr/bin/env python3
"""Quick GPU triage: runs a ladder of fwd/bwd shapes, prints error metrics vs a torch fp32
reference, one subprocess per stage so a faulting kernel does not hide the later stages."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

STAGES = [
    # name, b, sq, sk, h, hk, d, causal, dtype, bwd
    ("fwd_nc_512", 1, 512, 512, 4, 4, 128, False, "fp16", False),
    ("fwd_c_512", 1, 512, 512, 4, 4, 128, True, "fp16", False),
    ("fwd_tail", 2, 300, 389, 4, 2, 128, True, "fp16", False),
    ("fwd_d64", 2, 300, 389, 4, 2, 64, True, "fp16", False),
    ("fwd_bf16", 1, 512, 512, 4, 4, 128, False, "bf16", False),
    ("bwd_nc_512", 1, 512, 512, 4, 4, 128, False, "fp16", True),
    ("bwd_c_tail_gqa", 2, 300, 389, 4, 2, 128, True, "fp16", True),
    ("bwd_d64", 2, 300, 389, 4, 2, 64, True, "fp16", True),
    ("bwd_bf16", 1, 512, 512, 4, 4, 128, True, "bf16", True),
]


def run_stage(args):
    import torch
    import _util as U
    import flash_attn_turing as F

    name, b, sq, sk, h, hk, d, causal, dtype, bwd = args
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(0)
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    q = torch.randn(b, sq, h, d, generator=gen).to(dev, tdt)
    k = torch.randn(b, sk, hk, d, generator=gen).to(dev, tdt)
    v = torch.randn(b, sk, hk, d, generator=gen).to(dev, tdt)
    do = torch.randn(b, sq, h, d, generator=gen).to(dev, tdt)
    o, lse = F.fwd(q, k, v, causal)
    torch.cuda.synchronize()
    ref = U.torch_attention_ref(q, k, v, do if bwd else None, causal)
    m = U.error_metrics(o.float().cpu().numpy(), ref[0].cpu().numpy())
    print(f"{name:16s} O   max_abs {m['max_abs']:.2e} mean_abs {m['mean_abs']:.2e} mean_rel {m['mean_rel']:.2e}  "
          f"LSE max {float((lse - ref[1]).abs().max()):.2e}", flush=True)
    if bwd:
        dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
        torch.cuda.synchronize()
        for got, r, n in ((dq, ref[2], "dQ"), (dk, ref[3], "dK"), (dv, ref[4], "dV")):
            m = U.error_metrics(got.float().cpu().numpy(), r.cpu().numpy())
            print(f"{'':16s} {n:3s} max_abs {m['max_abs']:.2e} mean_abs {m['mean_abs']:.2e} mean_rel {m['mean_rel']:.2e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_stage(STAGES[int(sys.argv[1])])
    else:
        for i, s in enumerate(STAGES):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(i)], timeout=300)
            if r.returncode != 0:
                print(f"{s[0]:16s} FAILED rc={r.returncode}", flush=True)
