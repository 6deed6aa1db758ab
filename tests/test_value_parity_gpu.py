"""GPU: VALUE parity at BASELINE.json's full sizes (configs[1..4]), not properties.

The full problems are far too big for the CPU oracle, but every (batch, head) pair is an independent problem and the
attention contract restricts cleanly to sub-problems, so exact expectations for PARTS of the full-size result are cheap:

  (a) one head at full length in fp32 on the GPU (tests/_util.py:torch_attention_ref; 16k x 16k fp32 scores = 1 GiB):
      O, LSE (and dQ, dK, dV for configs[3]) of three (batch, head) pairs - first, last, one interior;
  (b) the C oracle (oracle/attn_oracle.c, the reference algorithm with its rounding points) on row blocks: the causal mask is
      bottom-right aligned (reference mask.h:172), so  oracle(q[m0:m0+256], k[:m0+256], v[:m0+256], causal)  reproduces exactly
      rows m0..m0+255 of the square causal problem, and non-causal rows depend on their own q rows only;
  (c) backward: given LSE and D = rowsum(dO * O) of the FULL problem, dK / dV of a 128-key block involve that block's K / V rows
      only (P_ij = exp(s_ij - LSE_i), dS_ij = P_ij (dP_ij - D_i)), and dQ of a row block its own rows only: the oracle's backward
      on (all queries x 128 keys) and on (256 queries x all keys) gives exact expectations for the first / interior / last blocks.
      The oracle's backward is fed the fp32 expectation's O (rounded to the output format) and LSE, never the kernel's own.

Mirrors reference test_flash_attn.py:352-386,551-554 (fwd + bwd against an independent attention) at the sizes reference
README.md:7-16 quotes.  Tolerances: tests/_util.py:assert_close incl. the reference's PLAIN bounds (sk >= 64)."""
import numpy as np
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu

FWD_CASES = {
    # name: (b, s, h, d, causal, dtype)
    "c2_fwd_4k": (4, 4096, 32, 128, False, "fp16"),            # BASELINE configs[1]
    "c3_fwd_16k_causal": (4, 16384, 32, 128, True, "fp16"),    # BASELINE configs[2] (the headline)
    "c5shard_fwd_16k": (4, 16384, 32, 128, False, "fp16"),     # one GPU's share of BASELINE configs[4]
    # head_dim 64 at the sizes the default policy hands to the 16x16x32 forward (fp16, round 4; reference README.md:12-18 quotes hdim 64 too)
    "d64_fwd_8k_fp16": (2, 8192, 8, 64, False, "fp16"),
    "d64_fwd_8k_causal_fp16": (2, 8192, 8, 64, True, "fp16"),
}
BWD_CASES = {
    "c4_fwdbwd_8k_bf16": (4, 8192, 32, 128, False, "bf16"),    # BASELINE configs[3]
    "c4_shape_causal_fp16": (4, 8192, 32, 128, True, "fp16"),  # same shape through the causal instances (not a BASELINE config)
}


def _pairs(b, h):
    return ((0, 0), (b - 1, h - 1), (1, 7))        # first, last, one interior (batch, head)


def _rand(gpu, shape, dtype, seed):
    gen = torch.Generator(device=gpu).manual_seed(seed)
    return torch.randn(*shape, device=gpu, dtype=U.torch_dtype(dtype), generator=gen)


def _np(t):
    return t.detach().float().cpu().numpy()


_cache = {}


def _full_problem(gpu, name, with_bwd):
    """the full-size HIP result of one BASELINE config (computed once per session)"""
    if name in _cache:
        return _cache[name]
    import flash_attn_turing as F

    b, s, h, d, causal, dtype = (BWD_CASES if with_bwd else FWD_CASES)[name]
    seed = 1000 + sorted(list(FWD_CASES) + list(BWD_CASES)).index(name) * 10
    q, k, v = (_rand(gpu, (b, s, h, d), dtype, seed + i) for i in range(3))
    o, lse = F.fwd(q, k, v, causal)
    r = dict(q=q, k=k, v=v, o=o, lse=lse)
    if with_bwd:
        r["do"] = _rand(gpu, (b, s, h, d), dtype, seed + 3)
        r["dq"], r["dk"], r["dv"] = F.bwd(q, k, v, o, lse, r["do"], causal)
    torch.cuda.synchronize()
    _cache.clear()                      # one full problem resident at a time (C3 / C5 are 2 GiB each with outputs)
    _cache[name] = r
    return r


@pytest.mark.parametrize("name", list(FWD_CASES))
def test_forward_values_full_heads_vs_fp32(gpu, name):
    """(a): O and LSE of three whole heads of the full-size problem against fp32 math."""
    b, s, h, d, causal, dtype = FWD_CASES[name]
    r = _full_problem(gpu, name, False)
    for bi, hi in _pairs(b, h):
        sl = (slice(bi, bi + 1), slice(None), slice(hi, hi + 1))
        o_ref, lse_ref = U.torch_attention_ref(r["q"][sl], r["k"][sl], r["v"][sl], None, causal)
        U.assert_close(_np(r["o"][sl]), _np(o_ref), dtype, f"O {name} b{bi} h{hi}", sk=s)
        dl = (r["lse"][bi, hi] - lse_ref[0, 0]).abs().max().item()
        assert dl <= U.LSE_TOL, f"LSE {name} b{bi} h{hi}: {dl}"
        del o_ref, lse_ref
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name", list(FWD_CASES))
def test_forward_values_row_blocks_vs_c_oracle(gpu, name):
    """(b): first / interior (straddling a 256-row workgroup boundary) / last 256 query rows against the C oracle."""
    from oracle import attn_oracle as A

    b, s, h, d, causal, dtype = FWD_CASES[name]
    mode = A.ROUND_FP16 if dtype == "fp16" else A.ROUND_BF16
    r = _full_problem(gpu, name, False)
    pairs = _pairs(b, h)
    for m0 in (0, s // 2 - 128, s - 256):
        rows = slice(m0, m0 + 256)
        keys = slice(0, m0 + 256) if causal else slice(None)
        # the three heads become one 3-head oracle problem (one host thread per head)
        qn = np.stack([_np(r["q"][bi, rows, hi]) for bi, hi in pairs], 1)[None]
        kn = np.stack([_np(r["k"][bi, keys, hi]) for bi, hi in pairs], 1)[None]
        vn = np.stack([_np(r["v"][bi, keys, hi]) for bi, hi in pairs], 1)[None]
        o_ref, lse_ref = A.attn_fwd(qn, kn, vn, causal=causal, round_mode=mode)
        got_o = np.stack([_np(r["o"][bi, rows, hi]) for bi, hi in pairs], 1)[None]
        got_l = np.stack([_np(r["lse"][bi, hi, rows]) for bi, hi in pairs], 0)[None]
        U.assert_close(got_o, o_ref, dtype, f"O {name} rows {m0}+256", sk=kn.shape[1])
        assert np.abs(got_l - lse_ref).max() <= U.LSE_TOL, f"LSE {name} rows {m0}+256"


@pytest.mark.parametrize("name", list(BWD_CASES))
def test_backward_values_full_heads_vs_fp32(gpu, name):
    """(a) for forward + backward: O, LSE, dQ, dK, dV of three whole heads against fp32 math + autograd."""
    b, s, h, d, causal, dtype = BWD_CASES[name]
    r = _full_problem(gpu, name, True)
    for bi, hi in _pairs(b, h):
        sl = (slice(bi, bi + 1), slice(None), slice(hi, hi + 1))
        o_ref, lse_ref, dq_ref, dk_ref, dv_ref = U.torch_attention_ref(r["q"][sl], r["k"][sl], r["v"][sl], r["do"][sl], causal)
        for got, ref, t in ((r["o"], o_ref, "O"), (r["dq"], dq_ref, "dQ"), (r["dk"], dk_ref, "dK"), (r["dv"], dv_ref, "dV")):
            U.assert_close(_np(got[sl]), _np(ref), dtype, f"{t} {name} b{bi} h{hi}", sk=s)
        dl = (r["lse"][bi, hi] - lse_ref[0, 0]).abs().max().item()
        assert dl <= U.LSE_TOL, f"LSE {name} b{bi} h{hi}: {dl}"
        del o_ref, lse_ref, dq_ref, dk_ref, dv_ref
        torch.cuda.empty_cache()


def test_backward_values_blocks_vs_c_oracle(gpu):
    """(c) at BASELINE configs[3] (bf16, non-causal): dK / dV of the first, an interior and the last 128-key block (= one dK/dV
    workgroup each) and dQ of the first / interior / last 256 query rows (= one dQ workgroup each) against the C oracle's backward,
    which is fed the fp32 expectation's O and LSE for the whole head."""
    from oracle import attn_oracle as A

    name = "c4_fwdbwd_8k_bf16"
    b, s, h, d, causal, dtype = BWD_CASES[name]
    r = _full_problem(gpu, name, True)
    pairs = _pairs(b, h)
    stack = lambda key, rows: np.stack([_np(r[key][bi, rows, hi]) for bi, hi in pairs], 1)[None]
    qn, kn, vn, don = (stack(n, slice(None)) for n in ("q", "k", "v", "do"))
    o_x, lse_x = [], []
    for bi, hi in pairs:
        sl = (slice(bi, bi + 1), slice(None), slice(hi, hi + 1))
        o_ref, lse_ref = U.torch_attention_ref(r["q"][sl], r["k"][sl], r["v"][sl], None, causal)
        o_x.append(U.round_like_output(_np(o_ref[0, :, 0]), dtype))
        lse_x.append(_np(lse_ref[0, 0]))
    on, lsen = np.stack(o_x, 1)[None], np.stack(lse_x, 0)[None]          # (1, s, 3, d), (1, 3, s)
    for n0 in (0, s // 2 - 64, s - 128):                                  # the interior one straddles two workgroups
        blk = slice(n0, n0 + 128)
        _, dk_ref, dv_ref = A.attn_bwd(qn, kn[:, blk], vn[:, blk], on, lsen, don, causal=False, round_mode=A.ROUND_BF16)
        U.assert_close(stack("dk", blk), dk_ref, dtype, f"dK {name} keys {n0}+128", sk=s)
        U.assert_close(stack("dv", blk), dv_ref, dtype, f"dV {name} keys {n0}+128", sk=s)
    for m0 in (0, s // 2 - 128, s - 256):
        rows = slice(m0, m0 + 256)
        dq_ref, _, _ = A.attn_bwd(qn[:, rows], kn, vn, on[:, rows], lsen[:, :, rows], don[:, rows], causal=False, round_mode=A.ROUND_BF16)
        U.assert_close(stack("dq", rows), dq_ref, dtype, f"dQ {name} rows {m0}+256", sk=s)
