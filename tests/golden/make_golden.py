#!/usr/bin/env python3
"""Generate the committed golden vectors from the REFERENCE's own Python oracles.

Dev-container only (needs /root/reference, never runs on the GPU box).  For every case the
inputs are seeded numpy normals rounded to fp16/bf16; expected outputs come from the
reference's `vanilla_attention_ref` (reference test_flash_attn.py:134-196) evaluated in fp32
on those rounded inputs (tighter than the fp16 oracle run the reference's tests use), cross
checked here against `memory_efficient_attention_ref` (:200-248).  LSE is not returned by
either reference function: it is computed as torch.logsumexp over the reference's masked
scores with -inf -> 0.0 (the kernel's dead-row convention, flash_fwd_kernel.h:767-771).
Varlen cases follow the reference's varlen test: the oracle is applied per sequence
(test_flash_attn.py:790-806).  Import recipe: SURVEY.md Appendix B.

Only DATA is written (inputs + expected outputs, .npz); no reference source is copied.
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference_tests():
    stub = types.ModuleType("flash_attn_turing")
    for n in ("fwd", "bwd", "varlen_fwd", "varlen_bwd"):
        setattr(stub, n, None)
    sys.modules["flash_attn_turing"] = stub
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import test_flash_attn as T
    return T


def lp_round(x, dtype):
    t = torch.from_numpy(x.astype(np.float32))
    return t.to(torch.float16 if dtype == "fp16" else torch.bfloat16)


def ref_lse(T, q, k, causal):
    """logsumexp of the reference's masked, scaled scores; (b, h, sq); dead rows -> 0."""
    qt = q.permute(0, 2, 1, 3).float()
    kt = k.permute(0, 2, 1, 3).float()
    ratio = qt.shape[1] // kt.shape[1]
    kt = kt.repeat_interleave(ratio, dim=1)
    s = torch.matmul(qt, kt.transpose(-2, -1)) / (q.shape[-1] ** 0.5)
    if causal:
        m = T.causal_lower_right(q.shape[1], k.shape[1], device=s.device)
        s = s.masked_fill(~m.view(1, 1, *m.shape), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    return torch.where(torch.isinf(lse), torch.zeros_like(lse), lse)


# name, b, sq, sk, h, hk, d, causal, dtype, row_subsample (store every n-th row of the outputs)
CASES = [
    ("c1_noncausal", 1, 512, 512, 4, 4, 128, False, "fp16", 8),   # BASELINE.json configs[0]
    ("c1_causal", 1, 512, 512, 4, 4, 128, True, "fp16", 8),       # same inputs as c1_noncausal (seed shared)
    ("mha_128", 1, 128, 128, 2, 2, 128, False, "fp16", 1),
    ("causal_256", 1, 256, 256, 2, 2, 128, True, "fp16", 2),
    ("tail_129_gqa", 1, 129, 129, 2, 1, 128, True, "fp16", 1),
    ("tail_63", 1, 63, 63, 2, 2, 128, False, "fp16", 1),
    ("sq64_sk256_causal", 1, 64, 256, 2, 1, 128, True, "fp16", 1),
    ("sq256_sk64_causal_deadrows", 1, 256, 64, 2, 2, 128, True, "fp16", 2),
    ("gqa_6_3", 1, 128, 128, 6, 3, 128, False, "fp16", 2),
    ("mqa_6_1_causal", 1, 128, 192, 6, 1, 128, True, "fp16", 2),
    ("d64_gqa_causal", 2, 128, 128, 4, 2, 64, True, "fp16", 2),
    ("d64_tail", 1, 65, 129, 2, 1, 64, False, "fp16", 1),
    ("bf16_256", 1, 256, 256, 2, 2, 128, False, "bf16", 2),
    ("bf16_causal_tail", 1, 200, 333, 2, 1, 128, True, "bf16", 2),
    ("tiny_1_1", 3, 1, 1, 2, 1, 128, True, "fp16", 1),
    ("tiny_2_1_causal", 1, 2, 1, 2, 1, 128, True, "fp16", 1),
    ("tiny_1_2", 1, 1, 2, 2, 2, 64, False, "fp16", 1),
    ("sq257_sk256_causal", 1, 257, 256, 2, 1, 128, True, "fp16", 4),
]

# varlen: name, seqlens_q, seqlens_k, h, hk, d, causal, dtype
VARLEN_CASES = [
    ("varlen_a", [37, 128, 5], [64, 100, 1], 2, 1, 128, True, "fp16"),
    ("varlen_b", [1, 100, 64, 17], [129, 64, 64, 2], 2, 2, 128, False, "fp16"),
    ("varlen_d64", [70, 1, 129], [70, 33, 64], 4, 1, 64, True, "fp16"),
]


def run_case(T, q, k, v, do, causal):
    qf, kf, vf, dof = q.float(), k.float(), v.float(), do.float()
    o, dq, dk, dv = T.vanilla_attention_ref(qf, kf, vf, dof, causal)
    o2, dq2, dk2, dv2 = T.memory_efficient_attention_ref(qf, kf, vf, dof, causal)
    for a, b_, n in ((o, o2, "o"), (dq, dq2, "dq"), (dk, dk2, "dk"), (dv, dv2, "dv")):
        # the SDPA oracle yields NaN for dead rows (softmax of all -inf); vanilla maps them to 0
        b_ = torch.nan_to_num(b_, nan=0.0)
        err = (a - b_).abs().max().item() if a.numel() else 0.0
        assert err < 5e-5, f"reference oracles disagree on {n}: {err}"
    return o.detach(), dq.detach(), dk.detach(), dv.detach()


def main():
    T = import_reference_tests()
    torch.set_num_threads(8)
    total = 0
    for idx, (name, b, sq, sk, h, hk, d, causal, dtype, sub) in enumerate(CASES):
        rng = np.random.default_rng(1000 + (0 if name.startswith('c1_') else idx))
        q = lp_round(rng.standard_normal((b, sq, h, d)), dtype)
        k = lp_round(rng.standard_normal((b, sk, hk, d)), dtype)
        v = lp_round(rng.standard_normal((b, sk, hk, d)), dtype)
        do = lp_round(rng.standard_normal((b, sq, h, d)), dtype)
        o, dq, dk, dv = run_case(T, q, k, v, do, causal)
        lse = ref_lse(T, q, k, causal)
        store = lambda t: t.view(torch.int16).numpy() if t.dtype == torch.bfloat16 else t.numpy()
        path = os.path.join(HERE, name + ".npz")
        inputs = {} if name == "c1_causal" else dict(q=store(q), k=store(k), v=store(v), dout=store(do))
        np.savez_compressed(
            path, **inputs,
            o=o.numpy()[:, ::sub], dq=dq.numpy()[:, ::sub], dk=dk.numpy()[:, ::sub], dv=dv.numpy()[:, ::sub],
            lse=lse.numpy()[:, :, ::sub],
            meta=np.array([b, sq, sk, h, hk, d, int(causal), 0 if dtype == "fp16" else 1, sub], dtype=np.int64))
        total += os.path.getsize(path)
        print(f"{name:32s} {os.path.getsize(path) / 1024:8.1f} KiB")
    for idx, (name, lq, lk, h, hk, d, causal, dtype) in enumerate(VARLEN_CASES):
        rng = np.random.default_rng(2000 + idx)
        q = lp_round(rng.standard_normal((sum(lq), h, d)), dtype)
        k = lp_round(rng.standard_normal((sum(lk), hk, d)), dtype)
        v = lp_round(rng.standard_normal((sum(lk), hk, d)), dtype)
        do = lp_round(rng.standard_normal((sum(lq), h, d)), dtype)
        cu_q = np.concatenate([[0], np.cumsum(lq)]).astype(np.int32)
        cu_k = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
        o = torch.zeros(q.shape); dq = torch.zeros(q.shape); dk = torch.zeros(k.shape); dv = torch.zeros(v.shape)
        lse = torch.zeros(len(lq), h, max(lq))
        for i in range(len(lq)):
            qs, ks = slice(cu_q[i], cu_q[i + 1]), slice(cu_k[i], cu_k[i + 1])
            oi, dqi, dki, dvi = run_case(T, q[qs][None], k[ks][None], v[ks][None], do[qs][None], causal)
            o[qs], dq[qs], dk[ks], dv[ks] = oi[0], dqi[0], dki[0], dvi[0]
            lse[i, :, : lq[i]] = ref_lse(T, q[qs][None], k[ks][None], causal)[0]
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(
            path, q=q.numpy(), k=k.numpy(), v=v.numpy(), dout=do.numpy(), cu_seqlens_q=cu_q, cu_seqlens_k=cu_k,
            o=o.numpy(), dq=dq.numpy(), dk=dk.numpy(), dv=dv.numpy(), lse=lse.numpy(),
            meta=np.array([len(lq), max(lq), max(lk), h, hk, d, int(causal), 0, 1], dtype=np.int64))
        total += os.path.getsize(path)
        print(f"{name:32s} {os.path.getsize(path) / 1024:8.1f} KiB")
    print(f"total {total / 2**20:.2f} MiB")


if __name__ == "__main__":
    main()
