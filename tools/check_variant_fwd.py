#!/usr/bin/env python3
"""Value check of an A/B build's FORWARD against fp32 math before it is timed: dense, ragged, sq != sk, dead rows, GQA and packed sequences through the
C-ABI of the library given by --lib (tools/build_variant.py), under a pinned kernel policy.  Prints the worst deviations per shape; exit status 1 if any
shape breaks the suite's tolerances (tests/_util.py).  Usage: check_variant_fwd.py --lib tools/abl/libfa_x.so [--policy 1] [--d 128]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import _util as U  # noqa: E402
from flash_attn_turing import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--policy", type=int, default=1)
ap.add_argument("--d", type=int, default=128)
a = ap.parse_args()
if a.lib:
    capi.LIBRARY_PATH = os.path.abspath(a.lib)
capi.set_kernel_policy(a.policy)
dev = torch.device("cuda:0")
# (b, sq, sk, h, hk, causal, dtype)
SHAPES = [(1, 512, 512, 4, 4, False, "fp16"), (1, 512, 512, 4, 4, True, "fp16"), (2, 300, 389, 4, 2, True, "fp16"), (2, 389, 300, 4, 2, True, "fp16"),
          (2, 1, 1, 2, 1, True, "fp16"), (1, 383, 33, 2, 2, False, "fp16"), (1, 385, 31, 2, 2, True, "fp16"), (1, 384, 384, 2, 2, True, "fp16"),
          (1, 769, 1025, 2, 1, True, "fp16"), (1, 1025, 769, 2, 1, True, "fp16"), (1, 2048, 2048, 4, 4, True, "fp16"), (1, 2048, 2048, 4, 4, False, "bf16"),
          (1, 4096, 4100, 2, 2, True, "bf16"), (2, 3000, 5000, 2, 1, False, "fp16"), (1, 2500, 640, 2, 2, True, "fp16"), (1, 8192, 8192, 2, 2, True, "fp16")]
bad = 0
for b, sq, sk, h, hk, causal, dt in SHAPES:
    tdt = torch.float16 if dt == "fp16" else torch.bfloat16
    gen = torch.Generator(device="cpu").manual_seed(sq * 31 + sk)
    q = torch.randn(b, sq, h, a.d, generator=gen).to(dev, tdt)
    k = torch.randn(b, sk, hk, a.d, generator=gen).to(dev, tdt)
    v = torch.randn(b, sk, hk, a.d, generator=gen).to(dev, tdt)
    o = torch.full_like(q, float("nan"))
    lse = torch.full((b, h, sq), float("nan"), device=dev, dtype=torch.float32)
    capi.mha_fwd(q, k, v, o, lse, causal)
    torch.cuda.synchronize()
    o_r, lse_r = U.torch_attention_ref(q, k, v, None, causal)
    m = U.error_metrics(o.float().cpu().numpy(), o_r.cpu().numpy())
    dl = float((lse - lse_r).abs().max())
    ok = True
    try:
        U.assert_close(o.float().cpu().numpy(), o_r.cpu().numpy(), dt, "O", sk=sk)
        assert dl <= U.LSE_TOL
    except AssertionError as e:
        ok, bad = False, bad + 1
        print("   ", str(e)[:200])
    print(f"{'ok ' if ok else 'BAD'} b{b} sq{sq} sk{sk} h{h}/{hk} causal={int(causal)} {dt}: O max_abs {m['max_abs']:.2e} mean_abs {m['mean_abs']:.2e}  LSE max {dl:.2e}", flush=True)

# packed sequences (compact grid, heavy-first lookup under a causal mask)
L = capi.lib()
for causal in (False, True):
    rng = np.random.default_rng(7 + causal)
    lq, lk = rng.integers(1, 1500, 7), rng.integers(1, 1500, 7)
    lq[2], lk[4] = 1500, 1500
    cu_q = np.concatenate([[0], np.cumsum(lq)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
    gen = torch.Generator(device="cpu").manual_seed(99)
    h, hk = 4, 2
    q = torch.randn(int(cu_q[-1]), h, a.d, generator=gen).to(dev, torch.float16)
    k = torch.randn(int(cu_k[-1]), hk, a.d, generator=gen).to(dev, torch.float16)
    v = torch.randn(int(cu_k[-1]), hk, a.d, generator=gen).to(dev, torch.float16)
    o = torch.full_like(q, float("nan"))
    lse = torch.zeros(7, h, 1500, device=dev, dtype=torch.float32)
    cq, ck = torch.from_numpy(cu_q).to(dev), torch.from_numpy(cu_k).to(dev)
    vp = ctypes.c_void_p
    rc = L.fa_mha_varlen_fwd(vp(q.data_ptr()), vp(k.data_ptr()), vp(v.data_ptr()), vp(o.data_ptr()), vp(lse.data_ptr()), vp(cq.data_ptr()), vp(ck.data_ptr()),
                             7, 1500, 1500, h, hk, a.d, 0, int(causal), vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, capi.last_error()
    torch.cuda.synchronize()
    worst, worst_l = 0.0, 0.0
    for i in range(7):
        qs, ks = slice(cu_q[i], cu_q[i + 1]), slice(cu_k[i], cu_k[i + 1])
        o_r, lse_r = U.torch_attention_ref(q[qs][None], k[ks][None], v[ks][None], None, causal)
        try:
            U.assert_close(o[qs].float().cpu().numpy(), o_r[0].cpu().numpy(), "fp16", f"O seq{i}", sk=int(lk[i]))
            assert float((lse[i, :, : lq[i]] - lse_r[0]).abs().max()) <= U.LSE_TOL
            assert bool((lse[i, :, lq[i]:] == 0).all())
        except AssertionError as e:
            bad += 1
            print("    varlen", str(e)[:200])
        worst = max(worst, float((o[qs].float() - o_r[0]).abs().max()))
        worst_l = max(worst_l, float((lse[i, :, : lq[i]] - lse_r[0]).abs().max()))
    print(f"varlen causal={int(causal)}: O max_abs {worst:.2e} LSE max {worst_l:.2e}", flush=True)
print("FAILED" if bad else "all shapes within tolerance")
sys.exit(1 if bad else 0)
