"""GPU: size-independent properties at BASELINE.json's full sizes (too big for the CPU oracle),
plus direct calls through the C ABI (ctypes, flat entry points)."""
import ctypes

import numpy as np
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu

C2 = (4, 4096, 32, 128)      # BASELINE configs[1]
C3 = (4, 16384, 32, 128)     # BASELINE configs[2]
C5 = (4, 16384, 32, 128)     # one GPU's share of BASELINE configs[4] (b=32 over 8 GPUs), NON-causal


def _rand(gpu, shape, dtype=torch.float16, seed=0):
    gen = torch.Generator(device=gpu).manual_seed(seed)
    return torch.randn(*shape, device=gpu, dtype=dtype, generator=gen)


@pytest.mark.parametrize("shape,causal", [(C2, False), (C3, True), (C5, False)])
def test_softmax_rows_sum_to_one_and_lse_consistent(gpu, shape, causal):
    """V = ones  =>  O = 1 exactly where a key is visible (P rows sum to 1 within fp16 rounding of P);
    and LSE must equal an fp32 logsumexp recomputed for sampled rows."""
    import flash_attn_turing as F

    b, s, h, d = shape
    q, k = _rand(gpu, (b, s, h, d), seed=1), _rand(gpu, (b, s, h, d), seed=2)
    o, lse = F.fwd(q, k, torch.ones_like(k), causal)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    assert (o.float() - 1.0).abs().max().item() <= 2e-3
    rows = torch.tensor([0, 1, 63, 64, 255, 256, 257, s // 2, s - 2, s - 1], device=gpu)
    for bi, hi in ((0, 0), (b - 1, h - 1), (1, 7)):
        sc = (q[bi, rows, hi].float() @ k[bi, :, hi].float().T) / d ** 0.5
        if causal:
            sc = sc.masked_fill(torch.arange(s, device=gpu)[None, :] > rows[:, None], float("-inf"))
        assert (torch.logsumexp(sc, -1) - lse[bi, hi, rows]).abs().max().item() <= U.LSE_TOL


def test_causal_first_rows_copy_v(gpu):
    """causal, sq == sk: query 0 sees only key 0 => O[0] == V[0] bit-exactly; query 1 is a 2-key blend."""
    import flash_attn_turing as F

    b, s, h, d = C3
    q, k, v = (_rand(gpu, (b, s, h, d), seed=i) for i in (3, 4, 5))
    o, lse = F.fwd(q, k, v, True)
    assert torch.equal(o[:, 0], v[:, 0])
    s0 = (q[:, 0].float() * k[:, 0].float()).sum(-1) / d ** 0.5
    assert (lse[:, :, 0] - s0).abs().max().item() <= 1e-4


@pytest.mark.parametrize("shape", [C2, C5], ids=["c2_4k", "c5shard_16k"])
def test_key_permutation_invariance_noncausal(gpu, shape):
    """softmax attention does not depend on the order of the keys (non-causal): permuting K and V rows
    changes tile membership and the online-softmax path, never the result (up to fp32 summation order).
    C5 = configs[4]'s per-GPU workload: 256 steady-state key tiles per workgroup, K/V of a head = 2x an XCD's L2."""
    import flash_attn_turing as F

    b, s, h, d = shape
    q, k, v = (_rand(gpu, (b, s, h, d), seed=i) for i in (6, 7, 8))
    perm = torch.randperm(s, device=gpu, generator=torch.Generator(device=gpu).manual_seed(9))
    o1, l1 = F.fwd(q, k, v, False)
    o2, l2 = F.fwd(q, k[:, perm].contiguous(), v[:, perm].contiguous(), False)
    assert (o1.float() - o2.float()).abs().max().item() <= 2e-3
    assert (l1 - l2).abs().max().item() <= 1e-4


def test_linearity_in_v_and_in_dout(gpu):
    """O is linear in V; dQ, dK, dV are linear in dO (fixed q, k, v)."""
    import flash_attn_turing as F

    b, s, h, d = 2, 4096, 16, 128
    q, k, v1, v2 = (_rand(gpu, (b, s, h, d), seed=i) for i in (10, 11, 12, 13))
    o1, _ = F.fwd(q, k, v1, True)
    o2, _ = F.fwd(q, k, v2, True)
    o12, _ = F.fwd(q, k, (v1.float() * 0.5 + v2.float() * 0.25).half(), True)
    assert (o12.float() - (0.5 * o1.float() + 0.25 * o2.float())).abs().max().item() <= 3e-3
    o, lse = F.fwd(q, k, v1, True)
    d1, d2 = _rand(gpu, (b, s, h, d), seed=14), _rand(gpu, (b, s, h, d), seed=15)
    g1 = F.bwd(q, k, v1, o, lse, d1, True)
    g2 = F.bwd(q, k, v1, o, lse, d2, True)
    g12 = F.bwd(q, k, v1, o, lse, (d1.float() * 0.5 + d2.float() * 0.5).half(), True)
    for a, b_, c_ in zip(g1, g2, g12):
        assert (c_.float() - 0.5 * (a.float() + b_.float())).abs().max().item() <= 6e-3


@pytest.mark.parametrize("hk", [4, 16])
@pytest.mark.parametrize("causal,s", [(True, 2048), (False, 2048), (False, 4096)])
def test_batch_head_sharding_equals_whole_and_is_deterministic(gpu, causal, s, hk):
    """The multi-GPU decomposition (independent (batch, head) problems, strided shard views) gives
    bit-identical results to the unsharded call - forward and backward, also at the sizes where head_dim 128 switches
    kernel sets: since round 6 the choice follows how far the LAUNCH fills the chip, so the shards run under `problem_policy(batch, heads)`, which states the
    whole problem's size to the policy (include/flash_attn_gfx950.h, fa_set_policy_problem_heads); the test also checks that the whole problem and a lone shard
    WOULD be served by different kernels somewhere on this grid (otherwise it proves nothing) - and repeated calls are bit-identical.  dK / dV of a GQA group are the one exception: how many workgroups share a group's query heads
    (the fp32 workspace split, C ABI 3) depends on how full the launch would leave the chip, so a shard may sum the same products in another
    order; there they agree to a rounding of the output format."""
    import flash_attn_turing as F

    b, h, d = 4, 16, 128
    q, do = _rand(gpu, (b, s, h, d), seed=16), _rand(gpu, (b, s, h, d), seed=19)
    k, v = _rand(gpu, (b, s, hk, d), seed=17), _rand(gpu, (b, s, hk, d), seed=18)
    o, lse = F.fwd(q, k, v, causal)
    o_again, lse_again = F.fwd(q, k, v, causal)
    assert torch.equal(o, o_again) and torch.equal(lse, lse_again)
    dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
    plans = F.plan_shards(b, h, hk, 8)
    sb, sh = plans[0].batch_stop - plans[0].batch_start, plans[0].head_stop - plans[0].head_start
    _SHARD_KERNELS_DIFFER.append(any(F.kernel_name(st, b, s, s, h, d, causal) != F.kernel_name(st, sb, s, s, sh, d, causal) for st in ("fwd", "dq")))
    with F.problem_policy(b, h):
        assert all(F.kernel_name(st, b, s, s, h, d, causal) == F.kernel_name(st, sb, s, s, sh, d, causal) for st in ("fwd", "dq", "dkdv"))
        _check_shards(F, plans, q, k, v, do, o, lse, dq, dk, dv, causal, hk == h)
    from flash_attn_turing import capi

    assert capi.set_policy_problem_heads(0) == 0          # restored


_SHARD_KERNELS_DIFFER = []


def test_sharding_grid_above_crosses_a_policy_boundary():
    """(runs after the parametrised test above) at least one of its cases must be one where a lone shard and the whole problem get different kernel sets"""
    assert _SHARD_KERNELS_DIFFER and any(_SHARD_KERNELS_DIFFER)


def _check_shards(F, plans, q, k, v, do, o, lse, dq, dk, dv, causal, mha):
    for plan in plans:
        qs, ks, vs, dos = F.shard_tensor(q, plan, False), F.shard_tensor(k, plan, True), F.shard_tensor(v, plan, True), F.shard_tensor(do, plan, False)
        os_, ls_ = F.fwd(qs, ks, vs, causal)
        assert torch.equal(os_, F.shard_tensor(o, plan, False))
        assert torch.equal(ls_, lse[plan.batch_start:plan.batch_stop, plan.head_start:plan.head_stop])
        dqs, dks, dvs = F.bwd(qs, ks, vs, os_, ls_, dos, causal)
        assert torch.equal(dqs, F.shard_tensor(dq, plan, False))
        if mha:
            assert torch.equal(dks, F.shard_tensor(dk, plan, True)) and torch.equal(dvs, F.shard_tensor(dv, plan, True))
        else:
            for got, whole in ((dks, F.shard_tensor(dk, plan, True)), (dvs, F.shard_tensor(dv, plan, True))):
                assert (got.float() - whole.float()).abs().max().item() <= 2 ** -9 * max(1.0, whole.float().abs().max().item())


def test_backward_full_size_c4_sanity(gpu):
    """BASELINE configs[3] shape (bf16, b4 s8192 h32 d128): finite, and dV column sums match the
    closed form  sum_j dV_j = sum_i dO_i  (every P row sums to 1)."""
    import flash_attn_turing as F

    b, s, h, d = 4, 8192, 32, 128
    q, k, v, do = (_rand(gpu, (b, s, h, d), torch.bfloat16, seed=i) for i in (19, 20, 21, 22))
    o, lse = F.fwd(q, k, v, False)
    dq, dk, dv = F.bwd(q, k, v, o, lse, do, False)
    for t in (dq, dk, dv):
        assert torch.isfinite(t).all()
    lhs, rhs = dv.float().sum(1), do.float().sum(1)
    assert ((lhs - rhs).abs().max() / rhs.abs().max()).item() <= 2e-2
    # softmax shift invariance: sum_j dS_ij = 0  =>  sum over keys of dK weighted ... check dQ.q - dK.k balance
    a1 = (dq.float() * q.float()).sum((1, 3))
    a2 = (dk.float() * k.float()).sum((1, 3))
    assert ((a1 - a2).abs().max() / a1.abs().max().clamp_min(1e-3)).item() <= 5e-2


def test_c_abi_flat_entry_points_match_host_module(gpu):
    """ctypes -> fa_mha_fwd / fa_mha_bwd / fa_mha_varlen_fwd with raw device pointers."""
    import flash_attn_turing as F
    from flash_attn_turing import capi

    b, sq, sk, h, hk, d = 2, 300, 389, 4, 2, 128
    q, do = _rand(gpu, (b, sq, h, d), seed=30), _rand(gpu, (b, sq, h, d), seed=31)
    k, v = _rand(gpu, (b, sk, hk, d), seed=32), _rand(gpu, (b, sk, hk, d), seed=33)
    o = torch.full_like(q, float("nan"))
    lse = torch.full((b, h, sq), float("nan"), device=gpu)
    capi.mha_fwd(q, k, v, o, lse, True)
    o_ref, lse_ref = F.fwd(q, k, v, True)
    torch.cuda.synchronize()
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    dsum = torch.empty_like(lse)
    capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, True)
    r = F.bwd(q, k, v, o, lse, do, True)
    torch.cuda.synchronize()
    assert torch.equal(dq, r[0])
    # GQA: the host module hands the dK/dV launch its fp32 scratch (ABI 3, head-group split), the flat entry point cannot; the two
    # sum the same fp32 terms in a different order and round once, so they agree to the last rounding, not to the bit
    for got, ref in ((dk, r[1]), (dv, r[2])):
        assert ((got.float() - ref.float()).abs() <= 2.0 ** -9 * ref.float().abs().clamp_min(1e-2)).all()
    assert (dsum - (o.float() * do.float()).sum(-1).permute(0, 2, 1)).abs().max().item() <= 1e-3
    # varlen through the flat entry point == fixed-length call on each sequence
    L = capi.lib()
    cu = torch.tensor([0, sq, 2 * sq], device=gpu, dtype=torch.int32)
    cuk = torch.tensor([0, sk, 2 * sk], device=gpu, dtype=torch.int32)
    qp, kp, vp = q.reshape(b * sq, h, d), k.reshape(b * sk, hk, d), v.reshape(b * sk, hk, d)
    op = torch.empty_like(qp)
    lp = torch.zeros(b, h, sq, device=gpu)
    s = torch.cuda.current_stream().cuda_stream
    rc = L.fa_mha_varlen_fwd(qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), op.data_ptr(), lp.data_ptr(), cu.data_ptr(), cuk.data_ptr(),
                             b, sq, sk, h, hk, d, 0, 1, s)
    assert rc == 0, capi.last_error()
    torch.cuda.synchronize()
    assert torch.equal(op.reshape(b, sq, h, d), o) and torch.equal(lp, lse)
    # a bad call reports an error code instead of launching
    assert L.fa_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), b, sq, sk, 3, 2, d, 0, 0, s) == capi.FA_ERR_BAD_GQA


def test_kernels_run_on_the_current_stream(gpu):
    """work is enqueued on torch's current stream (the reference uses the legacy default stream)."""
    import flash_attn_turing as F

    q, k, v = (_rand(gpu, (1, 512, 4, 128), seed=i) for i in (40, 41, 42))
    ref, _ = F.fwd(q, k, v, False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out, _ = F.fwd(q, k, v, False)
    side.synchronize()
    assert torch.equal(out, ref)


def test_64bit_offsets_at_config5_total_size(gpu):
    """BASELINE configs[4] total size on ONE GPU: b=32, s=16384, h=32, d=128 is exactly 2^31 elements per
    tensor -- the reference's int32 offsets (block_info.h:15-21) overflow here.  The last batch entries of
    the big call must be bit-identical to computing them on their own."""
    import flash_attn_turing as F

    b, s, h, d = 32, 16384, 32, 128
    gen = torch.Generator(device=gpu).manual_seed(50)
    q = torch.randn(b, s, h, d, device=gpu, dtype=torch.float16, generator=gen)
    k = torch.randn(b, s, h, d, device=gpu, dtype=torch.float16, generator=gen)
    v = torch.randn(b, s, h, d, device=gpu, dtype=torch.float16, generator=gen)
    assert q.numel() == 2 ** 31
    o, lse = F.fwd(q, k, v, True)
    for bi in (0, 15, 16, 31):                      # element offsets 0, just below / at 2^30, and near 2^31
        o1, l1 = F.fwd(q[bi:bi + 1], k[bi:bi + 1], v[bi:bi + 1], True)
        assert torch.equal(o[bi:bi + 1], o1) and torch.equal(lse[bi:bi + 1], l1), f"batch {bi}"
    del o, lse
    torch.cuda.empty_cache()


def test_empty_and_degenerate_shapes(gpu):
    import flash_attn_turing as F

    h = lambda *s: torch.randn(*s, device=gpu, dtype=torch.float16)
    # no keys at all: every row is dead -> O = 0, LSE = 0; gradients are zero
    q, k, v = h(2, 5, 4, 128), h(2, 0, 2, 128), h(2, 0, 2, 128)
    o, lse = F.fwd(q, k, v, False)
    assert o.shape == q.shape and (o == 0).all() and (lse == 0).all()
    dq, dk, dv = F.bwd(q, k, v, o, lse, torch.ones_like(q), False)
    assert (dq == 0).all() and dk.shape == k.shape and dv.shape == v.shape
    # no queries
    q0 = h(2, 0, 4, 128)
    k1, v1 = h(2, 7, 2, 128), h(2, 7, 2, 128)
    o0, l0 = F.fwd(q0, k1, v1, True)
    assert o0.shape == q0.shape and l0.shape == (2, 4, 0)
    dq0, dk0, dv0 = F.bwd(q0, k1, v1, o0, l0, q0.clone(), True)
    assert dq0.numel() == 0 and (dk0 == 0).all() and (dv0 == 0).all()
    # varlen with an empty sequence in the middle
    cu_q = torch.tensor([0, 3, 3, 10], device=gpu, dtype=torch.int32)
    cu_k = torch.tensor([0, 4, 9, 9], device=gpu, dtype=torch.int32)     # last sequence has queries but no keys
    qp, kp, vp = h(10, 2, 64), h(9, 2, 64), h(9, 2, 64)
    o, lse = F.varlen_fwd(qp, kp, vp, cu_q, cu_k, 7, 5, True)
    assert torch.isfinite(o).all() and (o[3:] == 0).all() and (lse[2] == 0).all()
    dq, dk, dv = F.varlen_bwd(qp, kp, vp, o, lse, torch.ones_like(qp), cu_q, cu_k, 7, 5, True)
    assert torch.isfinite(dq).all() and (dq[3:] == 0).all() and (dk[4:] == 0).all() and (dv[4:] == 0).all()


def test_large_magnitude_inputs_stay_finite(gpu):
    """scores of +-several hundred (raw q.k up to ~5000): exp2 arguments far outside fp16 range must not
    produce inf/NaN; the result must still match fp32 math."""
    import flash_attn_turing as F

    gen = torch.Generator(device="cpu").manual_seed(7)
    q = (torch.randn(1, 384, 2, 128, generator=gen) * 6).to(gpu, torch.float16)
    k = (torch.randn(1, 640, 2, 128, generator=gen) * 6).to(gpu, torch.float16)
    v = torch.randn(1, 640, 2, 128, generator=gen).to(gpu, torch.float16)
    for causal in (False, True):
        o, lse = F.fwd(q, k, v, causal)
        o_r, lse_r = U.torch_attention_ref(q, k, v, None, causal)
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        U.assert_close(o.float().cpu().numpy(), o_r.cpu().numpy(), "fp16", "O large", scale=2.0)
        assert ((lse - lse_r).abs() / lse_r.abs().clamp_min(1.0)).max().item() <= 1e-4


def test_launch_path_is_hip_graph_capturable(gpu):
    """The C ABI promises no allocation, no synchronisation and no host-dependent state at launch (include/flash_attn_gfx950.h), so
    forward + backward must be capturable in a HIP graph and replay to the same bits - dense (single-pass dK/dV) and varlen (compact grid:
    the slot lookup reads cu_seqlens on the device, nothing on the host; with a caller-provided workspace, i.e. the split dK/dV + sum kernel)."""
    import ctypes

    from flash_attn_turing import capi

    L = capi.lib()
    b, s, h, hk, d, dt = 3, 300, 4, 2, 128, torch.float16
    gen = torch.Generator(device="cpu").manual_seed(21)
    q, do = (torch.randn(b, s, h, d, generator=gen).to(gpu, dt) for _ in range(2))
    k, v = (torch.randn(b, s, hk, d, generator=gen).to(gpu, dt) for _ in range(2))
    o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lse = torch.empty(b, h, s, device=gpu, dtype=torch.float32)
    dsum = torch.empty_like(lse)

    def dense(stream):
        capi.mha_fwd(q, k, v, o, lse, True, stream=stream)
        capi.mha_bwd(q, k, v, o, lse, do, dq, dk, dv, dsum, True, stream=stream)

    # varlen through the param structs with totals set -> compact grid
    lens = [700, 0, 33, 64, 1]
    tot, nb, mx = sum(lens), len(lens), max(lens)
    qv, dov = (torch.randn(tot, h, d, generator=gen).to(gpu, dt) for _ in range(2))
    kv, vv = (torch.randn(tot, hk, d, generator=gen).to(gpu, dt) for _ in range(2))
    ov, dqv, dkv, dvv = torch.zeros_like(qv), torch.zeros_like(qv), torch.zeros_like(kv), torch.zeros_like(vv)
    lsev = torch.zeros(nb, h, mx, device=gpu, dtype=torch.float32)
    dsv = torch.zeros_like(lsev)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(gpu)
    row = lambda t: capi.Strides(0, t.stride(0), t.stride(1))
    common = dict(q=qv.data_ptr(), k=kv.data_ptr(), v=vv.data_ptr(), o=ov.data_ptr(), lse=lsev.data_ptr(), cu_seqlens_q=cu.data_ptr(), cu_seqlens_k=cu.data_ptr(),
                  b=nb, seqlen_q=mx, seqlen_k=mx, h=h, h_k=hk, d=d, dtype=capi.dtype_code(dt), is_causal=0,
                  q_stride=row(qv), k_stride=row(kv), v_stride=row(vv), o_stride=row(ov), total_q=tot, total_k=tot)
    fp = capi.FwdParams(**common)
    bp = capi.BwdParams(dout=dov.data_ptr(), dq=dqv.data_ptr(), dk=dkv.data_ptr(), dv=dvv.data_ptr(), dsoftmax_sum=dsv.data_ptr(),
                        do_stride=row(dov), dq_stride=row(dqv), dk_stride=row(dkv), dv_stride=row(dvv), **common)

    ws = capi.attach_workspace(bp, qv)            # ABI 3: GQA 4/2 on a small grid -> the dK/dV launch splits the head group (dK/dV + sum kernel)
    assert ws is not None and bp.workspace_bytes > 0

    def varlen(stream):
        capi.check(L.fa_run_mha_fwd(ctypes.byref(fp), stream))
        capi.check(L.fa_run_mha_bwd(ctypes.byref(bp), stream))

    outs = (o, lse, dq, dk, dv, ov, lsev, dqv, dkv, dvv)
    cur = torch.cuda.current_stream(gpu).cuda_stream
    dense(cur); varlen(cur)                       # eager reference run (also warms up lazy init before capture)
    torch.cuda.synchronize()
    ref = [t.clone() for t in outs]
    for t in outs:
        if t is lsev:
            t.zero_()        # padded varlen LSE entries are never written (the host module zero-fills them); valid entries are non-zero in ref
        elif t.dtype == torch.float32:
            t.fill_(123.0)
        else:
            t.fill_(float("nan"))
    assert not torch.equal(lsev, ref[6])
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            dense(side.cuda_stream); varlen(side.cuda_stream)
    torch.cuda.synchronize()
    for rep in range(2):                           # nothing ran during capture; replay twice
        graph.replay()
        torch.cuda.synchronize()
        for name, g, r in zip(("O", "LSE", "dQ", "dK", "dV", "O varlen", "LSE varlen", "dQ varlen", "dK varlen", "dV varlen"), outs, ref):
            assert torch.equal(g, r), f"{name} differs after graph replay {rep}"


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_identity_inputs_analytic_known_answer(gpu, d, causal):
    """Analytic known-answer test on the HIP kernels, independent of every reference implementation (the reference's identity-input
    debug mode, test_flash_attn.py:74-109: q and k one-hot with index row % d): s_ij = 1/sqrt(d) where (i - j) % d == 0 else 0, so
    P, O = P V and LSE have closed forms.  Dense and varlen (the packed batch restarts the index at every sequence), lengths above d
    so that the index wraps, sq != sk with the bottom-right causal alignment."""
    import flash_attn_turing as F

    h, hk = 4, 2
    rng = np.random.default_rng(31 + d)

    def onehot(n):
        t = np.zeros((n, d), np.float32)
        t[np.arange(n), np.arange(n) % d] = 1.0
        return t

    def expect(sq, sk, v):            # v: (sk, hk, d) float32 -> O (sq, h, d), LSE (h, sq) in float64
        i, j = np.arange(sq)[:, None], np.arange(sk)[None, :]
        s = np.where((i - j) % d == 0, 1.0 / np.sqrt(d), 0.0)
        if causal:
            s = np.where(j - i > sk - sq, -np.inf, s)
        m = s.max(-1, keepdims=True)
        dead = ~np.isfinite(m[:, 0])
        e = np.exp(s - np.where(np.isfinite(m), m, 0.0))
        l = e.sum(-1)
        p = np.where(dead[:, None], 0.0, e / np.where(l > 0, l, 1.0)[:, None])
        lse = np.where(dead, 0.0, np.log(np.where(l > 0, l, 1.0)) + np.where(np.isfinite(m[:, 0]), m[:, 0], 0.0))
        vv = np.repeat(v.astype(np.float64), h // hk, axis=1)           # (sk, h, d)
        return np.einsum("ij,jhd->ihd", p, vv), np.broadcast_to(lse, (h, sq))

    shapes = [(96, 96), (300, 300), (200, 333), (333, 200)]
    # dense, one shape at a time
    for sq, sk in shapes:
        q = torch.from_numpy(np.broadcast_to(onehot(sq)[None, :, None, :], (1, sq, h, d)).copy()).to(gpu, torch.float16)
        k = torch.from_numpy(np.broadcast_to(onehot(sk)[None, :, None, :], (1, sk, hk, d)).copy()).to(gpu, torch.float16)
        vn = rng.standard_normal((sk, hk, d)).astype(np.float16).astype(np.float32)
        v = torch.from_numpy(vn[None]).to(gpu, torch.float16)
        o, lse = F.fwd(q, k, v, causal)
        eo, el = expect(sq, sk, vn)
        U.assert_close(o[0].float().cpu().numpy(), eo, "fp16", f"O identity dense sq={sq} sk={sk}")
        assert np.abs(lse[0].cpu().numpy() - el).max() <= U.LSE_TOL, (sq, sk)
    # the same four problems as ONE packed varlen batch
    lq, lk = [s[0] for s in shapes], [s[1] for s in shapes]
    cu_q = np.concatenate([[0], np.cumsum(lq)]).astype(np.int32); cu_k = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
    qn = np.concatenate([np.broadcast_to(onehot(n)[:, None, :], (n, h, d)) for n in lq])
    kn = np.concatenate([np.broadcast_to(onehot(n)[:, None, :], (n, hk, d)) for n in lk])
    vn = rng.standard_normal((int(cu_k[-1]), hk, d)).astype(np.float16).astype(np.float32)
    q, k, v = (torch.from_numpy(np.ascontiguousarray(x)).to(gpu, torch.float16) for x in (qn, kn, vn))
    o, lse = F.varlen_fwd(q, k, v, torch.from_numpy(cu_q).to(gpu), torch.from_numpy(cu_k).to(gpu), max(lq), max(lk), causal)
    for i, (sq, sk) in enumerate(shapes):
        eo, el = expect(sq, sk, vn[cu_k[i]:cu_k[i + 1]])
        U.assert_close(o[cu_q[i]:cu_q[i + 1]].float().cpu().numpy(), eo, "fp16", f"O identity varlen seq {i}")
        assert np.abs(lse[i, :, :sq].cpu().numpy() - el).max() <= U.LSE_TOL, ("varlen", i)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("ramp", [1.0, 0.02, -1.0])
def test_running_max_rising_along_the_key_axis(gpu, d, dtype, ramp):
    """The steady-state softmax of the forward first exponentiates against the running max AS IT STANDS and only falls back to the
    exact max / rescale path when a lane's partial row sum exceeds 2^6 (fa_fwd_pp.hip).  Scores that keep rising along the key
    axis force that fallback again and again (ramp = nats per 64-key tile: 1.0 -> a factor e per tile, the 32-term partial sums
    pass 64 within a tile or two of every refresh; 0.02 -> the max creeps, a refresh every few dozen tiles); -1.0 never triggers
    it after tile 0.  All must match fp32 math, forward and backward (the backward recomputes P from the LSE the forward wrote).
    The ramp is spread over all d components so operands stay O(1): a ramp along one direction needs |k| ~ 200, and dQ = dS K
    with dS rounded to 16 bits (the reference's own contract) then cancels catastrophically in ANY implementation."""
    import flash_attn_turing as F

    dt = U.torch_dtype(dtype)
    b, s, h = 1, 1536, 2
    gen = torch.Generator(device="cpu").manual_seed(11)
    q = torch.randn(b, s, h, d, generator=gen) * 0.5 + 1.0
    k = torch.randn(b, s, h, d, generator=gen) * 0.5 + (ramp * torch.arange(s).float() / 64.0 / d ** 0.5).view(1, s, 1, 1)
    v = torch.randn(b, s, h, d, generator=gen)
    do = torch.randn(b, s, h, d, generator=gen)
    q, k, v, do = (x.to(gpu, dt) for x in (q, k, v, do))
    for causal in (False, True):
        o, lse = F.fwd(q, k, v, causal)
        dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
        o_r, lse_r, dq_r, dk_r, dv_r = U.torch_attention_ref(q, k, v, do, causal)
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        assert ((lse - lse_r).abs() / lse_r.abs().clamp_min(1.0)).max().item() <= 1e-4
        # a steep ramp leaves a handful of effective keys per row: the relative metric is bounded by what the reference ALGORITHM
        # (C oracle, contract mode) itself achieves against the same fp32 expectation (tests/_util.py:check_mean_rel)
        from oracle import attn_oracle as A

        mode = A.ROUND_FP16 if dtype == "fp16" else A.ROUND_BF16
        n = lambda t: t.float().cpu().numpy()
        oo, ol = A.attn_fwd(n(q), n(k), n(v), causal=causal, round_mode=mode)
        odq, odk, odv = A.attn_bwd(n(q), n(k), n(v), oo, ol, n(do), causal=causal, round_mode=mode)
        U.assert_close(o.float().cpu().numpy(), o_r.cpu().numpy(), dtype, f"O ramp {ramp}", scale=2.0, oracle=oo)
        for name, x, r, orc in (("dQ", dq, dq_r, odq), ("dK", dk, dk_r, odk), ("dV", dv, dv_r, odv)):
            assert torch.isfinite(x).all()
            U.assert_close(x.float().cpu().numpy(), r.cpu().numpy(), dtype, f"{name} ramp {ramp} causal {causal}", scale=4.0, oracle=orc)


def test_nonfinite_scores_propagate_like_fp32_math(gpu):
    """Inf / NaN in the inputs: the optimistic pass must hand such tiles to the exact path (its `sum <= 2^6` test fails for
    both) instead of looping or silently dropping them; rows without a non-finite score are unaffected."""
    import flash_attn_turing as F

    gen = torch.Generator(device="cpu").manual_seed(12)
    q = torch.randn(1, 512, 1, 128, generator=gen).to(gpu, torch.float16)
    k = torch.randn(1, 1024, 1, 128, generator=gen).to(gpu, torch.float16)
    v = torch.randn(1, 1024, 1, 128, generator=gen).to(gpu, torch.float16)
    q2 = q.clone()
    q2[0, 7, 0, 3] = float("nan")          # one query row sees NaN scores everywhere
    o, lse = F.fwd(q2, k, v, False)
    o_ref, lse_ref = F.fwd(q, k, v, False)
    assert torch.isnan(o[0, 7]).all() and torch.isnan(lse[0, 0, 7])
    keep = torch.ones(512, dtype=torch.bool, device=gpu); keep[7] = False
    assert torch.equal(o[0, keep], o_ref[0, keep]) and torch.equal(lse[0, 0, keep], lse_ref[0, 0, keep])


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("h,hk,d", [(8, 1, 128), (16, 2, 64), (32, 1, 128)])
def test_dkdv_head_group_split_matches_single_pass(gpu, h, hk, d, causal):
    """ABI 3: with workspace the dK/dV launch deals a KV head's query heads to several workgroups (fp32 partial sums + a fixed-order
    sum kernel).  Same problem with and without workspace: both within tolerance of fp32 math, equal to each other up to the last
    rounding (fp32 summation order differs), the split result bit-reproducible, and a too-small workspace degrades to a smaller
    split instead of failing."""
    from flash_attn_turing import capi

    dt = torch.bfloat16 if d == 128 else torch.float16
    b, s = 2, 1024
    q, k, v, do = _rand(gpu, (b, s, h, d), dt, 1), _rand(gpu, (b, s, hk, d), dt, 2), _rand(gpu, (b, s, hk, d), dt, 3), _rand(gpu, (b, s, h, d), dt, 4)
    o = torch.empty_like(q)
    lse = torch.empty(b, h, s, device=gpu, dtype=torch.float32)
    capi.mha_fwd(q, k, v, o, lse, causal)
    outs = {}
    for mode in ("none", "full", "full_again", "half"):
        dq, dk, dv = torch.empty_like(q), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
        dsum = torch.empty_like(lse)
        p = capi.bwd_params(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
        need = capi.bwd_workspace_bytes(p)
        assert need > 0 and need % (2 * b * s * hk * d * 4) == 0, need        # whole planes: 2 tensors x n_split x rows x h_k x d fp32
        if mode != "none":
            n = need if mode != "half" else need // 2
            ws = torch.empty(n // 4, device=gpu, dtype=torch.float32)
            p.workspace, p.workspace_bytes = ws.data_ptr(), n
        capi.check(capi.lib().fa_run_mha_bwd(ctypes.byref(p), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        outs[mode] = (dq, dk, dv)
    _, _, dq_r, dk_r, dv_r = U.torch_attention_ref(q, k, v, do, causal)
    name = "bf16" if dt == torch.bfloat16 else "fp16"
    for mode, (dq, dk, dv) in outs.items():
        for nm, x, r in (("dQ", dq, dq_r), ("dK", dk, dk_r), ("dV", dv, dv_r)):
            U.assert_close(x.float().cpu().numpy(), r.cpu().numpy(), name, f"{nm} {mode}", scale=float(h // hk) ** 0.5, sk=s)
    for i in (1, 2):
        assert torch.equal(outs["full"][i], outs["full_again"][i]), "split result must be deterministic"
        ulp = 2.0 ** (-7 if dt == torch.bfloat16 else -10)
        for mode in ("full", "half"):
            diff = (outs[mode][i].float() - outs["none"][i].float()).abs()
            assert (diff <= 2 * ulp * outs["none"][i].float().abs().clamp_min(1e-2)).all(), f"{mode} vs single pass"


def test_dkdv_split_on_packed_sequences_with_padding_rows(gpu):
    """varlen + MQA: the split sizes its planes by total_k (= k.size(0), here 64 rows MORE than the tokens present); every sequence's
    gradients must still land on its own rows"""
    import flash_attn_turing as F

    h, hk, d = 8, 1, 128
    lens = [300, 1, 517, 129]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=gpu)
    total = int(cu[-1])
    q, do = _rand(gpu, (total, h, d), torch.float16, 5), _rand(gpu, (total, h, d), torch.float16, 6)
    k, v = _rand(gpu, (total + 64, hk, d), torch.float16, 7), _rand(gpu, (total + 64, hk, d), torch.float16, 8)   # 64 padding rows
    o, lse = F.varlen_fwd(q, k, v, cu, cu, max(lens), max(lens), True)
    dq, dk, dv = F.varlen_bwd(q, k, v, o, lse, do, cu, cu, max(lens), max(lens), True)
    for i, n in enumerate(lens):
        a, e = int(cu[i]), int(cu[i + 1])
        _, _, dq_r, dk_r, dv_r = U.torch_attention_ref(q[a:e][None], k[a:e][None], v[a:e][None], do[a:e][None], True)
        for nm, x, r in (("dQ", dq[a:e], dq_r[0]), ("dK", dk[a:e], dk_r[0]), ("dV", dv[a:e], dv_r[0])):
            U.assert_close(x.float().cpu().numpy(), r.cpu().numpy(), "fp16", f"{nm} seq {i}", scale=3.0)
