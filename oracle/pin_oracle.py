#!/usr/bin/env python3
"""Pin the C oracle against the REFERENCE's own Python oracles, directly (dev container only).

Imports /root/reference/test_flash_attn.py (recipe: SURVEY.md Appendix B) and, over a grid of shapes
in the spirit of the reference's parametrisation (test_flash_attn.py:251-343), checks
    oracle_attn_fwd / oracle_attn_bwd (round_mode NONE)   vs   vanilla_attention_ref  (fp32, :134-196)
                                                          and  memory_efficient_attention_ref (:200-248)
to 3e-5 (O) / 2e-4 (grads), and the rounded mode (the kernel contract) against the reference's
tolerances (:407-414).  Never runs on the GPU box and is not imported by the product path.
Usage: PYTHONDONTWRITEBYTECODE=1 python oracle/pin_oracle.py  (writes oracle/PIN_RESULTS.txt)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import attn_oracle as A  # noqa: E402


def import_reference_tests():
    stub = types.ModuleType("flash_attn_turing")
    for n in ("fwd", "bwd", "varlen_fwd", "varlen_bwd"):
        setattr(stub, n, None)
    sys.modules["flash_attn_turing"] = stub
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    import test_flash_attn as T
    return T


GRID = [(b, h, hk, sq, sk, d, c)
        for d in (64, 128) for (h, hk) in ((2, 1), (4, 2), (6, 3), (6, 1), (4, 4)) for b in (1, 3)
        for (sq, sk) in ((64, 64), (63, 63), (65, 129), (128, 257), (257, 64), (1, 1), (1, 2), (2, 1), (64, 1), (1, 128), (129, 65))
        for c in (False, True)]


def main():
    T = import_reference_tests()
    torch.set_num_threads(8)
    worst = dict(o=0.0, dq=0.0, dk=0.0, dv=0.0, o_sdpa=0.0, o_rounded_max_abs=0.0, grads_rounded_max_abs=0.0)
    for idx, (b, h, hk, sq, sk, d, causal) in enumerate(GRID):
        rng = np.random.default_rng(idx)
        f16 = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).half().float()
        q, k, v, do = f16(b, sq, h, d), f16(b, sk, hk, d), f16(b, sk, hk, d), f16(b, sq, h, d)
        o_r, dq_r, dk_r, dv_r = (t.detach().numpy() for t in T.vanilla_attention_ref(q, k, v, do, causal))
        o_s = torch.nan_to_num(T.memory_efficient_attention_ref(q, k, v, None, causal), nan=0.0).detach().numpy()
        o, lse = A.attn_fwd(q.numpy(), k.numpy(), v.numpy(), causal=causal, round_mode=A.ROUND_NONE)
        dq, dk, dv = A.attn_bwd(q.numpy(), k.numpy(), v.numpy(), o, lse, do.numpy(), causal=causal, round_mode=A.ROUND_NONE)
        for name, got, ref in (("o", o, o_r), ("dq", dq, dq_r), ("dk", dk, dk_r), ("dv", dv, dv_r), ("o_sdpa", o, o_s)):
            worst[name] = max(worst[name], float(np.abs(got - ref).max(initial=0)))
        assert worst["o"] <= 3e-5 and worst["o_sdpa"] <= 3e-5, (idx, worst)
        assert max(worst["dq"], worst["dk"], worst["dv"]) <= 2e-4, (idx, worst)
        # contract mode: P / dS / outputs rounded to fp16 like the reference kernels
        o16, lse16 = A.attn_fwd(q.numpy(), k.numpy(), v.numpy(), causal=causal, round_mode=A.ROUND_FP16)
        g16 = A.attn_bwd(q.numpy(), k.numpy(), v.numpy(), o16, lse16, do.numpy(), causal=causal, round_mode=A.ROUND_FP16)
        worst["o_rounded_max_abs"] = max(worst["o_rounded_max_abs"], float(np.abs(o16 - o_r).max(initial=0)))
        for got, ref in zip(g16, (dq_r, dk_r, dv_r)):
            err = np.abs(got - A.round_lp(ref, A.ROUND_FP16))
            scale = max(1.0, float(np.abs(ref).max(initial=0)))            # one output ulp at the tensor's magnitude
            worst["grads_rounded_max_abs"] = max(worst["grads_rounded_max_abs"], float(err.max(initial=0)) / scale)
    lines = [f"cases checked: {len(GRID)} (d in {{64,128}}, 5 head pairs incl. MHA, b in {{1,3}}, 11 (sq,sk) pairs, causal F/T)",
             "exact mode (round_mode NONE) max |C oracle - reference vanilla_attention_ref|:",
             *(f"  {k:6s} {worst[k]:.3e}" for k in ("o", "dq", "dk", "dv")),
             f"  O vs memory_efficient_attention_ref (SDPA): {worst['o_sdpa']:.3e}",
             "contract mode (P, dS, outputs rounded to fp16) vs the fp32 reference:",
             f"  O max_abs {worst['o_rounded_max_abs']:.3e}   grads max_abs / max(1,|ref|max) {worst['grads_rounded_max_abs']:.3e}   (reference tolerance 5e-3)"]
    assert worst["o_rounded_max_abs"] <= 5e-3 and worst["grads_rounded_max_abs"] <= 5e-3
    text = "\n".join(lines)
    print(text)
    with open(os.path.join(HERE, "PIN_RESULTS.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
