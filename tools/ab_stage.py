#!/usr/bin/env python3
"""Interleaved in-process A/B of individual kernels between builds of the library: forward, dQ and dK/dV each timed alone
through the stage-level C-ABI entry points (fa_run_mha_fwd / fa_bwd_dq / fa_bwd_dkdv).  Builds without the stage entry points
(round 1) are timed on the whole backward only.  Usage: ab_stage.py A.so B.so [...] [--only substr] [--stages fwd,dq,dkdv,bwd]"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi  # noqa: E402  (struct definitions only)

F16, BF16 = torch.float16, torch.bfloat16
CONFIGS = {
    "c4 bf16 d128 8k": (4, 8192, 32, 32, 128, BF16, False),
    "bf16 d128 8k causal": (4, 8192, 32, 32, 128, BF16, True),
    "c3 fp16 d128 16k causal": (4, 16384, 32, 32, 128, F16, True),
    "c5shard fp16 d128 16k": (4, 16384, 32, 32, 128, F16, False),
    "c2 fp16 d128 4k": (4, 4096, 32, 32, 128, F16, False),
    "fp16 d128 2k": (4, 2048, 32, 32, 128, F16, False),
    "fp16 d128 1k": (4, 1024, 32, 32, 128, F16, False),
    "fp16 d128 512": (4, 512, 32, 32, 128, F16, False),
    "fp16 d128 1k causal": (4, 1024, 32, 32, 128, F16, True),
    "fp16 d128 2k causal": (4, 2048, 32, 32, 128, F16, True),
    "fp16 d128 4k causal": (4, 4096, 32, 32, 128, F16, True),
    "fp16 d128 3k causal b2": (2, 3072, 32, 32, 128, F16, True),
    "fp16 d128 2k causal b16": (16, 2048, 32, 32, 128, F16, True),
    "fp16 d128 1k causal b32": (32, 1024, 32, 32, 128, F16, True),
    "fp16 d128 4k causal b8": (8, 4096, 32, 32, 128, F16, True),
    "fp16 d64 2k causal b16": (16, 2048, 32, 32, 64, F16, True),
    "fp16 d64 4k causal": (4, 4096, 32, 32, 64, F16, True),
    "fp16 d64 16k causal": (4, 16384, 32, 32, 64, F16, True),
    "bf16 d64 8k causal": (4, 8192, 32, 32, 64, BF16, True),
    "fp16 d64 2k causal": (4, 2048, 32, 32, 64, F16, True),
    "fp16 d128 512 causal": (4, 512, 32, 32, 128, F16, True),
    "fp16 d64 512": (4, 512, 32, 32, 64, F16, False),
    "fp16 d64 1k": (4, 1024, 32, 32, 64, F16, False),
    "bf16 d64 1k": (4, 1024, 32, 32, 64, BF16, False),
    "bf16 d64 2k": (4, 2048, 32, 32, 64, BF16, False),
    "fp16 d64 1k causal": (4, 1024, 32, 32, 64, F16, True),
    "fp16 d64 8k causal b1": (1, 8192, 32, 32, 64, F16, True),
    "fp16 d64 8k b1": (1, 8192, 32, 32, 64, F16, False),
    "fp16 d64 4k gqa": (4, 4096, 32, 8, 64, F16, False),
    "fp16 d64 8k gqa causal": (4, 8192, 32, 8, 64, F16, True),
    "fp16 d64 8k": (4, 8192, 32, 32, 64, F16, False),
    "fp16 d64 2k": (4, 2048, 32, 32, 64, F16, False),
    "fp16 d64 4k": (4, 4096, 32, 32, 64, F16, False),
    "fp16 d64 16k": (4, 16384, 32, 32, 64, F16, False),
    "bf16 d64 8k": (4, 8192, 32, 32, 64, BF16, False),
    "fp16 d64 8k causal": (4, 8192, 32, 32, 64, F16, True),
    "bf16 d128 8k mqa causal": (4, 8192, 32, 1, 128, BF16, True),
    "fp16 d128 gqa 4k causal": (4, 4096, 32, 8, 128, F16, True),
    "bf16 d128 8k mha causal b1": (1, 8192, 32, 32, 128, BF16, True),
    "bf16 d128 8k mqa causal b1": (1, 8192, 32, 1, 128, BF16, True),
    "bf16 d128 8k gqa4 causal": (4, 8192, 32, 4, 128, BF16, True),
    "bf16 d128 2k mqa": (1, 2048, 32, 1, 128, BF16, False),
    "bf16 d128 2k": (4, 2048, 32, 32, 128, BF16, False),
    "bf16 d128 4k": (4, 4096, 32, 32, 128, BF16, False),
    "bf16 d128 16k": (2, 16384, 32, 32, 128, BF16, False),
    "bf16 d128 8k gqa4": (4, 8192, 32, 8, 128, BF16, False),
    "fp16 d128 sq8k sk1k causal": (4, 8192, 32, 32, 128, F16, True, 1024),
    "fp16 d128 sq16k sk2k causal": (2, 16384, 32, 32, 128, F16, True, 2048),
    "fp16 d64 sq8k sk1k causal": (4, 8192, 32, 32, 64, F16, True, 1024),
    "fp16 d128 sq1k sk8k causal": (4, 1024, 32, 32, 128, F16, True, 8192),
}


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    vp = ctypes.c_void_p
    L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), vp]
    L.fa_run_mha_bwd.argtypes = [ctypes.POINTER(capi.BwdParams), vp]
    for n in ("fa_bwd_dot_do_o", "fa_bwd_dq", "fa_bwd_dkdv"):
        if hasattr(L, n):
            getattr(L, n).argtypes = [ctypes.POINTER(capi.BwdParams), vp]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--stages", default="fwd,dq,dkdv")
    ap.add_argument("--policy", type=int, default=None, help="fa_set_kernel_policy on every build that has it (0 = 32x32x16 set, 1 = 16x16x32 set, 2 = by size)")
    ap.add_argument("--no-workspace", action="store_true", help="do not give ABI 3 builds the dK/dV scratch (single-pass dK/dV)")
    a = ap.parse_args()
    libs = {f"{chr(65 + i)}:{os.path.basename(p)[6:-3]}": load(p) for i, p in enumerate(a.libs)}
    stages = a.stages.split(",")
    if a.policy is not None:
        for L in libs.values():
            if hasattr(L, "fa_set_kernel_policy"):
                L.fa_set_kernel_policy(a.policy)
    dev = torch.device("cuda:0")
    for cname, cfg in CONFIGS.items():
        if a.only and not any(x in cname for x in a.only.split(",")):
            continue
        b, s, h, hk, d, dt, causal = cfg[:7]
        sk = cfg[7] if len(cfg) > 7 else s          # (optional 8th entry: seqlen_k != seqlen_q)
        gen = torch.Generator(device=dev).manual_seed(1)
        q, do = (torch.randn(b, s, h, d, device=dev, dtype=dt, generator=gen) for _ in range(2))
        k, v = (torch.randn(b, sk, hk, d, device=dev, dtype=dt, generator=gen) for _ in range(2))
        o, dq = torch.empty_like(q), torch.empty_like(q)
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        lse, dsum = (torch.empty(b, h, s, device=dev, dtype=torch.float32) for _ in range(2))
        pf = capi.fwd_params(q, k, v, o, lse, causal)
        pb = capi.bwd_params(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
        # ABI 3 builds get the dK/dV scratch they ask for (head-group split for GQA / MQA); older builds ignore the trailing fields
        pbs, keep = {}, []
        for n, L in libs.items():
            pbs[n] = pb
            if hasattr(L, "fa_bwd_workspace_bytes") and not a.no_workspace:
                L.fa_bwd_workspace_bytes.restype = ctypes.c_int64
                L.fa_bwd_workspace_bytes.argtypes = [ctypes.POINTER(capi.BwdParams)]
                need = L.fa_bwd_workspace_bytes(ctypes.byref(pb))
                if need > 0:
                    pw = capi.bwd_params(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
                    ws = torch.empty(need // 4, device=dev, dtype=torch.float32)
                    pw.workspace, pw.workspace_bytes = ws.data_ptr(), need
                    pbs[n] = pw
                    keep.append(ws)
        st = torch.cuda.current_stream(dev).cuda_stream
        first = list(libs.values())[0]
        assert first.fa_run_mha_fwd(ctypes.byref(pf), st) == 0
        assert first.fa_run_mha_bwd(ctypes.byref(pb), st) == 0      # o, lse, dsum valid for every stage of every build
        torch.cuda.synchronize()
        ref = {}
        for stage in stages:
            pairs = s * sk * (0.5 if causal else 1.0) if sk == s else (sum(max(0, min(sk, i + sk - s + 1)) for i in range(s)) if causal else s * sk)
            fl = 4.0 * b * h * pairs * d * {"fwd": 1, "dq": 1.5, "dkdv": 2, "bwd": 2.5}[stage]
            times, outs = {n: [] for n in libs}, {}

            def run(n):
                L = libs[n]
                if stage == "fwd":
                    return L.fa_run_mha_fwd(ctypes.byref(pf), st)
                if stage == "bwd" or not hasattr(L, "fa_bwd_dq"):
                    return L.fa_run_mha_bwd(ctypes.byref(pbs[n]), st)
                return getattr(L, "fa_bwd_" + stage)(ctypes.byref(pbs[n]), st)
            for n in libs:
                assert run(n) == 0
                torch.cuda.synchronize()
                outs[n] = [t.clone() for t in ((o, lse) if stage == "fwd" else (dq,) if stage == "dq" else (dk, dv) if stage == "dkdv" else (dq, dk, dv))]
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
                whole = stage in ("dq", "dkdv") and not hasattr(libs[n], "fa_bwd_dq")
                same = (not whole) and all(torch.equal(x, y) for x, y in zip(outs[na], outs[n])) if not (stage in ("dq", "dkdv") and not hasattr(libs[na], "fa_bwd_dq")) else None
                dmax = max((x.float() - y.float()).abs().max().item() for x, y in zip(outs[na], outs[n])) if (same is False) else 0.0
                print(f"{cname:26s} {stage:5s} {n:28s} {med:8.3f} ms (min {min(ts):8.3f}) {fl / med / 1e9:6.0f} TF{' [whole bwd]' if whole else ''}  vs A {med / ma:6.4f}  same bits: {same}"
                      + (f"  max|diff| {dmax:.2e}" if same is False else ""), flush=True)


if __name__ == "__main__":
    main()
