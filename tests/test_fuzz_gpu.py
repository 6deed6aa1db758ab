"""GPU: deterministic randomised parity sweep over odd shapes (forward + backward against the plain PyTorch fp32 statement of the
contract, tolerances of tests/_util.py).  The reference's grid (test_attention_gpu.py) is regular; this one draws lengths from
tile-boundary neighbourhoods, primes and skewed sq / sk ratios, random GQA ratios, both head dims and dtypes, and varlen batches
that contain empty sequences.  Seeds are fixed: a failure reproduces from the case id alone."""
import numpy as np
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu

# FA_FUZZ_OFFSET=n shifts every seed of this file by n: the committed suite is offset 0; other offsets are for one-off wider sweeps
# (profiles/r2_fuzz_extra_offsets.log holds the last one)
import os  # noqa: E402
_OFF = int(os.environ.get("FA_FUZZ_OFFSET", "0"))

_EDGES = [1, 2, 3, 7, 16, 31, 32, 33, 63, 64, 65, 95, 97, 127, 128, 129, 191, 193, 255, 256, 257, 383, 389, 511, 512, 513, 769, 1021, 1031, 1279, 1543]


def _length(rng):
    r = rng.random()
    if r < 0.45:
        return int(rng.choice(_EDGES))
    if r < 0.75:
        return int(rng.integers(1, 80))
    return int(rng.integers(80, 1400))


def _heads(rng):
    h = int(rng.choice([1, 2, 3, 4, 6, 8]))
    hk = int(rng.choice([x for x in (1, 2, 3, 4, 6, 8) if h % x == 0]))
    return h, hk


def _check(got, ref, dt, tag, sk=None):
    for g, r, name in zip(got, ref, ("O", "dQ", "dK", "dV")):
        U.assert_close(g.float().cpu().numpy(), r.cpu().numpy(), dt, f"{name} {tag}", sk=sk)


@pytest.mark.parametrize("case", range(48))
def test_dense_random_shapes(gpu, case):
    import flash_attn_turing as F

    rng = np.random.default_rng(1000 + case + 100000 * _OFF)
    b = int(rng.integers(1, 4))
    sq, sk = _length(rng), _length(rng)
    if case % 6 == 0:
        sq, sk = max(sq, 700), min(sk, 40)       # sq >> sk (causal: most rows see no key)
    if case % 6 == 1:
        sq, sk = min(sq, 40), max(sk, 900)       # sk >> sq
    h, hk = _heads(rng)
    d = int(rng.choice([64, 128]))
    dt = ("fp16", "bf16")[case % 2]
    causal = bool(rng.integers(0, 2))
    tdt = U.torch_dtype(dt)
    gen = torch.Generator(device="cpu").manual_seed(7000 + case + 100000 * _OFF)
    q = torch.randn(b, sq, h, d, generator=gen).to(gpu, tdt)
    k = torch.randn(b, sk, hk, d, generator=gen).to(gpu, tdt)
    v = torch.randn(b, sk, hk, d, generator=gen).to(gpu, tdt)
    do = torch.randn(b, sq, h, d, generator=gen).to(gpu, tdt)
    tag = f"[case {case}: b{b} sq{sq} sk{sk} h{h}/{hk} d{d} {dt} causal={causal}]"
    if sq * sk <= 256 * 257:
        # the suite's rule for small problems (test_reference_grid_vs_torch_fp32): expectation = the CPU oracle WITH the reference's
        # rounding points (P, dS rounded to the 16-bit format before the second GEMMs); with a handful of keys per row nothing averages
        # that rounding out, and e.g. sk = 3, sq = 300, 4 q-heads per kv-head puts ~1 output ulp (std) of it on every dK element
        from oracle import attn_oracle as A

        mode = A.ROUND_FP16 if dt == "fp16" else A.ROUND_BF16
        n = lambda t: t.float().cpu().numpy()
        o_n, lse_n = A.attn_fwd(n(q), n(k), n(v), causal=causal, round_mode=mode)
        dq_n, dk_n, dv_n = A.attn_bwd(n(q), n(k), n(v), o_n, lse_n, n(do), causal=causal, round_mode=mode)
        o_r, lse_r, dq_r, dk_r, dv_r = (torch.from_numpy(x).to(gpu) for x in (o_n, lse_n, dq_n, dk_n, dv_n))
        tag += " vs C oracle"
    else:
        o_r, lse_r, dq_r, dk_r, dv_r = U.torch_attention_ref(q, k, v, do, causal)
    o, lse = F.fwd(q, k, v, causal)
    dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
    _check((o, dq, dk, dv), (o_r, dq_r, dk_r, dv_r), dt, tag, sk=sk)
    assert (lse - lse_r).abs().max().item() <= U.LSE_TOL, "LSE " + tag
    for t, name in ((o, "O"), (dq, "dQ"), (dk, "dK"), (dv, "dV"), (lse, "LSE")):
        assert torch.isfinite(t.float()).all(), f"non-finite {name} {tag}"


@pytest.mark.parametrize("case", range(16))
def test_varlen_random_batches_with_empty_sequences(gpu, case):
    import flash_attn_turing as F

    rng = np.random.default_rng(2000 + case + 100000 * _OFF)
    batch = int(rng.integers(1, 7))
    lq = np.array([_length(rng) if rng.random() > 0.2 else 0 for _ in range(batch)])
    lk = np.array([_length(rng) if rng.random() > 0.2 else 0 for _ in range(batch)])
    if lq.sum() == 0:
        lq[0] = 5
    if lk.sum() == 0:
        lk[0] = 9
    max_q, max_k = int(lq.max()), int(lk.max())
    h, hk = _heads(rng)
    d = int(rng.choice([64, 128]))
    dt = ("fp16", "bf16")[case % 2]
    causal = bool(rng.integers(0, 2))
    tdt = U.torch_dtype(dt)
    cu_q = np.concatenate([[0], np.cumsum(lq)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
    gen = torch.Generator(device="cpu").manual_seed(9000 + case + 100000 * _OFF)
    q = torch.randn(int(cu_q[-1]), h, d, generator=gen).to(gpu, tdt)
    k = torch.randn(int(cu_k[-1]), hk, d, generator=gen).to(gpu, tdt)
    v = torch.randn(int(cu_k[-1]), hk, d, generator=gen).to(gpu, tdt)
    do = torch.randn(int(cu_q[-1]), h, d, generator=gen).to(gpu, tdt)
    cq, ck = torch.from_numpy(cu_q).to(gpu), torch.from_numpy(cu_k).to(gpu)
    o, lse = F.varlen_fwd(q, k, v, cq, ck, max_q, max_k, causal)
    dq, dk, dv = F.varlen_bwd(q, k, v, o, lse, do, cq, ck, max_q, max_k, causal)
    for t, name in ((o, "O"), (dq, "dQ"), (dk, "dK"), (dv, "dV"), (lse, "LSE")):
        assert torch.isfinite(t.float()).all(), f"non-finite {name} case {case}"
    for i in range(batch):
        qs, ks = slice(cu_q[i], cu_q[i + 1]), slice(cu_k[i], cu_k[i + 1])
        tag = f"[case {case} seq {i}: lq={lq[i]} lk={lk[i]} h{h}/{hk} d{d} {dt} causal={causal}]"
        if lq[i] == 0:
            # no query rows: this sequence's keys receive no gradient
            assert (dk[ks] == 0).all() and (dv[ks] == 0).all(), "dK/dV of a sequence without queries must be zero " + tag
            continue
        if lk[i] == 0:
            # no keys: every row is dead -> O = 0, LSE = 0, dQ = 0 (the kernel's dead-row convention)
            assert (o[qs] == 0).all() and (dq[qs] == 0).all() and (lse[i, :, : lq[i]] == 0).all(), "rows without keys " + tag
            continue
        if int(lq[i]) * int(lk[i]) <= 256 * 257:     # same small-problem rule as above
            from oracle import attn_oracle as A

            mode = A.ROUND_FP16 if dt == "fp16" else A.ROUND_BF16
            n = lambda t: t[None].float().cpu().numpy()
            o_n, lse_n = A.attn_fwd(n(q[qs]), n(k[ks]), n(v[ks]), causal=causal, round_mode=mode)
            dq_n, dk_n, dv_n = A.attn_bwd(n(q[qs]), n(k[ks]), n(v[ks]), o_n, lse_n, n(do[qs]), causal=causal, round_mode=mode)
            o_r, lse_r, dq_r, dk_r, dv_r = (torch.from_numpy(x).to(gpu) for x in (o_n, lse_n, dq_n, dk_n, dv_n))
            tag += " vs C oracle"
        else:
            o_r, lse_r, dq_r, dk_r, dv_r = U.torch_attention_ref(q[qs][None], k[ks][None], v[ks][None], do[qs][None], causal)
        _check((o[qs], dq[qs], dk[ks], dv[ks]), (o_r[0], dq_r[0], dk_r[0], dv_r[0]), dt, tag, sk=int(lk[i]))
        assert (lse[i, :, : lq[i]] - lse_r[0]).abs().max().item() <= U.LSE_TOL, "LSE " + tag
        assert (lse[i, :, lq[i]:] == 0).all(), "padded LSE must stay zero " + tag


_LARGE = [(2049, 2049), (3001, 4099), (4099, 3001), (5000, 8191), (8191, 5000), (9001, 9001), (1, 8191), (8191, 1), (4099, 63), (63, 4099), (6007, 2050), (2050, 6007)]


@pytest.mark.parametrize("case", range(len(_LARGE) * 2))
def test_dense_large_odd_shapes(gpu, case):
    """many-tile loops with ragged tails: odd lengths of several thousand, GQA, both dtypes / head dims, causal and not"""
    import flash_attn_turing as F

    sq, sk = _LARGE[case // 2]
    causal = bool(case % 2)
    rng = np.random.default_rng(3000 + case)
    h, hk = _heads(rng)
    d = int(rng.choice([64, 128]))
    dt = ("fp16", "bf16")[(case // 2) % 2]
    tdt = U.torch_dtype(dt)
    gen = torch.Generator(device="cpu").manual_seed(11000 + case)
    q = torch.randn(1, sq, h, d, generator=gen).to(gpu, tdt)
    k = torch.randn(1, sk, hk, d, generator=gen).to(gpu, tdt)
    v = torch.randn(1, sk, hk, d, generator=gen).to(gpu, tdt)
    do = torch.randn(1, sq, h, d, generator=gen).to(gpu, tdt)
    tag = f"[large case {case}: sq{sq} sk{sk} h{h}/{hk} d{d} {dt} causal={causal}]"
    if sq * sk <= 256 * 257:
        from oracle import attn_oracle as A

        mode = A.ROUND_FP16 if dt == "fp16" else A.ROUND_BF16
        n = lambda t: t.float().cpu().numpy()
        o_n, lse_n = A.attn_fwd(n(q), n(k), n(v), causal=causal, round_mode=mode)
        dq_n, dk_n, dv_n = A.attn_bwd(n(q), n(k), n(v), o_n, lse_n, n(do), causal=causal, round_mode=mode)
        o_r, lse_r, dq_r, dk_r, dv_r = (torch.from_numpy(x).to(gpu) for x in (o_n, lse_n, dq_n, dk_n, dv_n))
        tag += " vs C oracle"
    else:
        o_r, lse_r, dq_r, dk_r, dv_r = U.torch_attention_ref(q, k, v, do, causal)
    o, lse = F.fwd(q, k, v, causal)
    dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
    _check((o, dq, dk, dv), (o_r, dq_r, dk_r, dv_r), dt, tag, sk=sk)
    assert (lse - lse_r).abs().max().item() <= U.LSE_TOL, "LSE " + tag


def _layouts(b, s_q, s_k, h, hk, d, tdt, gpu, gen):
    """name -> (q, k, v, dout) views over differently laid-out storage, all with the same VALUES as the contiguous base set"""
    base = {n: torch.randn(*shape, generator=gen).to(gpu, tdt) for n, shape in
            (("q", (b, s_q, h, d)), ("k", (b, s_k, hk, d)), ("v", (b, s_k, hk, d)), ("dout", (b, s_q, h, d)))}

    def bhsd(t):          # (b, h, s, d) storage viewed as (b, s, h, d): the layout torch SDPA users hold
        return t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)

    def sbhd(t):          # (s, b, h, d) storage
        return t.permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3)

    def padded(t, ps, ph, pd, off):
        bb, ss, hh, dd = t.shape
        buf = torch.zeros(bb, ss + ps + off, hh + ph, dd + pd, device=t.device, dtype=t.dtype)
        view = buf[:, off:off + ss, :hh, :dd]
        view.copy_(t)
        return view

    out = {"contiguous": tuple(base[n] for n in ("q", "k", "v", "dout")),
           "bhsd storage": tuple(bhsd(base[n]) for n in ("q", "k", "v", "dout")),
           "sbhd storage": tuple(sbhd(base[n]) for n in ("q", "k", "v", "dout")),
           "padded, aligned (row/head pitch d+8, 1 row in)": tuple(padded(base[n], 3, 1, 8, 1) for n in ("q", "k", "v", "dout")),
           "padded, UNALIGNED head pitch d+4 (wrapper must copy)": tuple(padded(base[n], 0, 0, 4, 0) for n in ("q", "k", "v", "dout"))}
    kv = torch.stack((base["k"], base["v"]), dim=2)                      # (b, s_k, 2, hk, d) packed KV
    out["packed kv"] = (base["q"], kv[:, :, 0], kv[:, :, 1], base["dout"])
    if h == hk and s_q == s_k:
        qkv = torch.stack((base["q"], base["k"], base["v"]), dim=2)       # (b, s, 3, h, d) packed QKV
        out["packed qkv"] = (qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], base["dout"])
    return out


@pytest.mark.parametrize("case", range(8))
def test_strided_views_are_bit_identical_to_contiguous(gpu, case):
    """the kernels take real strides: any accepted layout must give exactly the contiguous result, forward and backward"""
    import flash_attn_turing as F

    rng = np.random.default_rng(4000 + case)
    b = int(rng.integers(1, 4))
    s_q = int(rng.choice([1, 63, 128, 257, 700]))
    s_k = s_q if case % 2 == 0 else int(rng.choice([1, 65, 300, 513]))
    h, hk = _heads(rng)
    if case % 2 == 0:
        hk = h                                 # exercise packed qkv
    d = int(rng.choice([64, 128]))
    dt = ("fp16", "bf16")[case % 2]
    causal = bool(rng.integers(0, 2))
    gen = torch.Generator(device="cpu").manual_seed(13000 + case)
    lay = _layouts(b, s_q, s_k, h, hk, d, U.torch_dtype(dt), gpu, gen)
    ref = None
    for name, (q, k, v, do) in lay.items():
        o, lse = F.fwd(q, k, v, causal)
        dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
        got = (o, lse, dq, dk, dv)
        if ref is None:
            ref = got
            continue
        for g, r, tname in zip(got, ref, ("O", "LSE", "dQ", "dK", "dV")):
            assert torch.equal(g, r), f"{tname} differs for layout '{name}' [case {case}: b{b} sq{s_q} sk{s_k} h{h}/{hk} d{d} {dt} causal={causal}] " \
                                      f"max|diff| {(g.float() - r.float()).abs().max().item():.3e}"


@pytest.mark.parametrize("case", range(11))
def test_varlen_compact_grid_is_bit_identical_to_plain_grid(gpu, case):
    """fa_*_params.total_q / total_k only change WHICH workgroup computes a tile (grid sized by the tokens present instead of
    max_seqlen x batch, fa_device.hpp): results through the param-struct entry points must not change by a bit, skewed batches with
    empty sequences included; with totals unset the same call uses the plain grid."""
    import ctypes

    from flash_attn_turing import capi

    rng = np.random.default_rng(5000 + case)
    if case < 6:
        lens = [int(rng.choice([1500, 2600, 4100]))] + [int(rng.choice([0, 1, 17, 64, 130, 300])) for _ in range(int(rng.integers(5, 40)))]
    else:
        # more than 64 sequences: every lane of the slot lookup owns SEVERAL sequences (2 at 70-128, 3 at 130, 5 at 300, 8 at 512);
        # 513 is past the lookup's capacity and must fall back to the plain grid
        nseq = {6: 70, 7: 130, 8: 300, 9: 512, 10: 513}[case]
        lens = [int(rng.choice([700, 1300]))] + [int(rng.choice([0, 1, 5, 33, 64, 100])) for _ in range(nseq - 1)]
    rng.shuffle(lens)
    h, hk = _heads(rng)
    d = int(rng.choice([64, 128]))
    dt = ("fp16", "bf16")[case % 2]
    causal = bool(case % 3 == 0)
    _compact_vs_plain(gpu, lens, h, hk, d, dt, causal, 15000 + case, f"case {case}")


@pytest.mark.parametrize("name,lens,h,hk", [
    ("long first", [2048] * 3 + [0, 1, 40, 300, 64, 129] * 5, 8, 8),
    ("long last, 64 sequences", [17, 0, 256, 257, 1, 90] * 10 + [1500, 0, 2600, 700], 16, 4),
    ("65 sequences: one per lane no longer fits, the sequence-major lookup serves", [33] * 60 + [1500, 0, 2600, 700, 5], 8, 2),
    ("one sequence of exactly 64 query tiles = 128 key blocks (dK/dV falls back, forward and dQ do not)", [16384, 300, 0, 64], 8, 8),
    ("65 query tiles: fallback for every kernel", [16400, 300], 8, 8),
    ("equal lengths", [512] * 12, 24, 24),
    ("a single sequence", [1000], 8, 8),
])
@pytest.mark.parametrize("d", [128, 64])
def test_varlen_causal_heavy_first_lookup_is_bit_identical_to_plain_grid(gpu, name, lens, h, hk, d):
    """Round 4: a CAUSAL packed batch on the compact grid is dispatched heaviest items first across sequences and heads
    (fa_device.hpp:varlen_slot_lookup_heavy_first: heads a multiple of 8, at most 64 sequences, at most 64 tiles per sequence).  Only the order in
    which workgroups find their (sequence, tile) changes: every output must equal the plain grid's bit for bit, on both sides of each limit."""
    _compact_vs_plain(gpu, lens, h, hk, d, "fp16" if d == 128 else "bf16", True, 777 + len(lens), name)


def _compact_vs_plain(gpu, lens, h, hk, d, dt, causal, seed, tag):
    import ctypes

    from flash_attn_turing import capi

    tdt = U.torch_dtype(dt)
    tot, b, mx = sum(lens), len(lens), max(lens)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    q, do = (torch.randn(tot, h, d, generator=gen).to(gpu, tdt) for _ in range(2))
    k, v = (torch.randn(tot, hk, d, generator=gen).to(gpu, tdt) for _ in range(2))
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(gpu)
    L = capi.lib()
    st = torch.cuda.current_stream(gpu).cuda_stream
    row = lambda t: capi.Strides(0, t.stride(0), t.stride(1))
    res = {}
    for total in (0, tot, tot + 1000):          # unknown -> plain grid; exact; an upper bound (slack slots must exit cleanly)
        o, dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        lse = torch.zeros(b, h, mx, device=gpu, dtype=torch.float32)
        dsum = torch.zeros_like(lse)
        common = dict(q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=o.data_ptr(), lse=lse.data_ptr(), cu_seqlens_q=cu.data_ptr(), cu_seqlens_k=cu.data_ptr(),
                      b=b, seqlen_q=mx, seqlen_k=mx, h=h, h_k=hk, d=d, dtype=capi.dtype_code(tdt), is_causal=int(causal),
                      q_stride=row(q), k_stride=row(k), v_stride=row(v), o_stride=row(o), total_q=total, total_k=total)
        fp = capi.FwdParams(**common)
        bp = capi.BwdParams(dout=do.data_ptr(), dq=dq.data_ptr(), dk=dk.data_ptr(), dv=dv.data_ptr(), dsoftmax_sum=dsum.data_ptr(),
                            do_stride=row(do), dq_stride=row(dq), dk_stride=row(dk), dv_stride=row(dv), **common)
        capi.check(L.fa_run_mha_fwd(ctypes.byref(fp), st))
        capi.check(L.fa_run_mha_bwd(ctypes.byref(bp), st))
        torch.cuda.synchronize()
        res[total] = (o, lse, dq, dk, dv)
    for total in (tot, tot + 1000):
        for g, r, name in zip(res[total], res[0], ("O", "LSE", "dQ", "dK", "dV")):
            assert torch.equal(g, r), f"{name} differs between compact (total={total}) and plain grid [{tag}: lens {lens} h{h}/{hk} d{d} {dt} causal={causal}]"
