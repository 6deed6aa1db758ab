"""CPU: the C oracle (oracle/attn_oracle.c) against the committed golden vectors, which were
produced by the REFERENCE's own Python oracles (tests/golden/make_golden.py)."""
import numpy as np
import pytest

import _util as U
from oracle import attn_oracle as A


@pytest.mark.parametrize("name", U.golden_names())
def test_oracle_exact_math_matches_reference_vectors(name):
    """round_mode NONE = the reference oracle's own math (fp32 softmax attention): tight match."""
    g = U.load_golden(name)
    kw = dict(causal=g["causal"], round_mode=A.ROUND_NONE)
    if g["varlen"]:
        kw.update(cu_seqlens_q=g["cu_seqlens_q"], cu_seqlens_k=g["cu_seqlens_k"], max_seqlen_q=g["sq"], max_seqlen_k=g["sk"])
    o, lse = A.attn_fwd(g["q"], g["k"], g["v"], **kw)
    dq, dk, dv = A.attn_bwd(g["q"], g["k"], g["v"], o, lse, g["dout"], **kw)
    o, dq, dk, dv, lse = U.subsample(g, o, dq, dk, dv, lse)
    assert np.abs(o - g["o"]).max(initial=0) <= 2e-5
    assert np.abs(lse - g["lse"]).max(initial=0) <= 2e-5
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert np.abs(got - g[key]).max(initial=0) <= 1e-4, key


@pytest.mark.parametrize("name", U.golden_names())
def test_oracle_with_reference_rounding_points_within_reference_tolerance(name):
    """round_mode fp16/bf16 = the kernel contract (P, dS, outputs rounded): reference tolerances."""
    g = U.load_golden(name)
    mode = A.ROUND_FP16 if g["dtype"] == "fp16" else A.ROUND_BF16
    kw = dict(causal=g["causal"], round_mode=mode)
    if g["varlen"]:
        kw.update(cu_seqlens_q=g["cu_seqlens_q"], cu_seqlens_k=g["cu_seqlens_k"], max_seqlen_q=g["sq"], max_seqlen_k=g["sk"])
    o, lse = A.attn_fwd(g["q"], g["k"], g["v"], **kw)
    dq, dk, dv = A.attn_bwd(g["q"], g["k"], g["v"], o, lse, g["dout"], **kw)
    o, dq, dk, dv, lse = U.subsample(g, o, dq, dk, dv, lse)
    U.assert_close(o, g["o"], g["dtype"], "O")
    U.assert_close(dq, g["dq"], g["dtype"], "dQ")
    U.assert_close(dk, g["dk"], g["dtype"], "dK")
    U.assert_close(dv, g["dv"], g["dtype"], "dV")
    assert np.abs(lse - g["lse"]).max(initial=0) <= 2e-5


def test_dead_rows_and_lse_convention():
    g = U.load_golden("sq256_sk64_causal_deadrows")
    o, lse = A.attn_fwd(g["q"], g["k"], g["v"], causal=True, round_mode=A.ROUND_FP16)
    n_dead = g["sq"] - g["sk"]                      # rows i with i + sk - sq < 0
    assert np.all(o[:, :n_dead] == 0.0) and np.all(lse[:, :, :n_dead] == 0.0)
    assert np.all(np.abs(o[:, n_dead:]).sum(-1) > 0)


def test_identity_inputs_known_answer():
    """Analytic KAT in the spirit of the reference's identity-input debug mode
    (reference test_flash_attn.py:74-109): one-hot rows, index = row % d."""
    sq = sk = 96
    d, h = 64, 2
    eye = np.zeros((1, sq, h, d), np.float32)
    eye[0, np.arange(sq), :, np.arange(sq) % d] = 1.0
    v = np.random.default_rng(0).standard_normal((1, sk, h, d)).astype(np.float16).astype(np.float32)
    o, lse = A.attn_fwd(eye, eye, v, causal=False, round_mode=A.ROUND_NONE)
    # score = 1/sqrt(d) where (i - j) % d == 0 else 0
    s = np.where((np.arange(sq)[:, None] - np.arange(sk)[None, :]) % d == 0, 1.0 / np.sqrt(d), 0.0)
    p = np.exp(s) / np.exp(s).sum(-1, keepdims=True)
    expect = np.einsum("ij,jhd->ihd", p, v[0])
    assert np.abs(o[0] - expect).max() < 1e-5
    assert np.abs(lse[0, 0] - np.log(np.exp(s).sum(-1))).max() < 1e-5


def test_dot_do_o():
    g = U.load_golden("mha_128")
    o, _ = A.attn_fwd(g["q"], g["k"], g["v"], round_mode=A.ROUND_FP16)
    dsum = A.dot_do_o(o, g["dout"])
    expect = np.einsum("bshd,bshd->bhs", o.astype(np.float64), g["dout"].astype(np.float64))
    assert np.abs(dsum - expect).max() < 1e-5
