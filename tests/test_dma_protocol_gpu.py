"""GPU: the LDS-DMA protocols of fa_fwd_pp16 (role-split rings, counted waits) and fa_bwd_dkdv16 (toggled Q / dO rings) under ADVERSARIAL timing.

Why this module exists.  A race of the class "a consumer reads a ring slot before the wait + barrier that publishes the producer's
pieces" is invisible to every value test as long as the DMA is usually early: round 4's first counted-wait form was WRONG and bit-identical on every
shape (profiles/r4_fwd_counted_wait_racy_form_ab.log; found by walking the two wave groups' timelines side by side).  Determinism soaks cannot see it
either.  What can: builds of the SAME kernels whose DMA requests are moved, mechanically, to the latest point the protocol itself allows - directly in
front of the wait that retires them (`dma_late`), or behind a sleep longer than a tile period in one wave group (`dma_sleepy`).  A correct protocol
does not care when the bytes land as long as its own waits and barriers are honoured, so both builds must reproduce the product BIT FOR BIT; a protocol
that reads before the publishing barrier reads the slot's previous tenant and fails.  To show the method has teeth, the documented racy form is kept
in the source behind a test-only switch (`dma_racy`: bit-identical at normal timing, as round 4 recorded) and must FAIL under late issue
(`dma_racy_late`).  The debug libraries are built by flash-attention-turing_amd/build.py (DEBUG_VARIANTS) next to the product; the product's ISA is unchanged."""
import ctypes
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, b, sq, sk, h, hk, dtype, causal): every shape reaches the unrolled steady loop of fa_fwd_pp16 under the default policy
# (asserted through capi.kernel_name in the test body - the policy's thresholds are stated in include/flash_attn_gfx950.h only; >= 4 unmasked tiles per workgroup) and fa_bwd_dkdv16
SHAPES = [
    ("c2_b4_s4096_h32", 4, 4096, 4096, 32, 32, torch.float16, False),            # BASELINE configs[1]
    ("causal_b1_s8192_h16", 1, 8192, 8192, 16, 16, torch.float16, True),
    ("ragged_b2_sq5000_sk5100_gqa", 2, 5000, 5100, 8, 2, torch.float16, False),
    ("bf16_b2_s4096_h8", 2, 4096, 4096, 8, 8, torch.bfloat16, False),
    # head_dim 64: the round-5 instances of both kernels - 128-key forward tiles, dK/dV with one piece per wave and ring and the 2^13 ring-toggle bit - share the switches
    ("d64_fp16_b2_s4096_h8", 2, 4096, 4096, 8, 8, torch.float16, False, 64),
    ("d64_fp16_causal_b1_s8192_h8", 1, 8192, 8192, 8, 8, torch.float16, True, 64),
]


def _build_module():
    spec = importlib.util.spec_from_file_location("fa_build", os.path.join(ROOT, "flash-attention-turing_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def libs(gpu):
    from flash_attn_turing import capi

    fa_build = _build_module()
    fa_build.build_debug_variants(force=False)          # a no-op when build() has run (the .so files travel with the tree); hipcc is on the GPU box otherwise
    out = {"product": capi.lib()}
    for name in fa_build.DEBUG_VARIANTS:
        L = ctypes.CDLL(fa_build.debug_library_path(name))
        L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
        L.fa_run_mha_bwd.argtypes = [ctypes.POINTER(capi.BwdParams), ctypes.c_void_p]
        L.fa_kernel_name_dtype.restype = ctypes.c_char_p
        L.fa_kernel_name_dtype.argtypes = [ctypes.c_int32] * 8
        out[name] = L
    return out


def _inputs(gpu, b, sq, sk, h, hk, dt, d=128):
    g = torch.Generator(device=gpu).manual_seed(20260929)
    q, do = (torch.randn(b, sq, h, d, device=gpu, dtype=dt, generator=g) for _ in range(2))
    k, v = (torch.randn(b, sk, hk, d, device=gpu, dtype=dt, generator=g) for _ in range(2))
    return q, k, v, do


def _fwd(L, q, k, v, causal):
    from flash_attn_turing import capi

    o = torch.full_like(q, float("nan"))
    lse = torch.full((q.shape[0], q.shape[2], q.shape[1]), float("nan"), device=q.device, dtype=torch.float32)
    p = capi.fwd_params(q, k, v, o, lse, causal)
    assert L.fa_run_mha_fwd(ctypes.byref(p), torch.cuda.current_stream(q.device).cuda_stream) == 0
    torch.cuda.synchronize()
    return o, lse


def _bwd(L, q, k, v, o, lse, do, causal):
    from flash_attn_turing import capi

    dq, dk, dv = (torch.full_like(t, float("nan")) for t in (q, k, v))
    dsum = torch.empty_like(lse)
    p = capi.bwd_params(q, k, v, o, lse, do, dq, dk, dv, dsum, causal)
    assert L.fa_run_mha_bwd(ctypes.byref(p), torch.cuda.current_stream(q.device).cuda_stream) == 0
    torch.cuda.synchronize()
    return dq, dk, dv


@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_product_protocol_is_bit_identical_under_adversarial_dma_timing(gpu, libs, shape):
    from flash_attn_turing import capi

    _, b, sq, sk, h, hk, dt, causal = shape[:8]
    d = shape[8] if len(shape) > 8 else 128
    dtn = "fp16" if dt == torch.float16 else "bf16"
    assert capi.kernel_name("fwd", b, sq, sk, h, d, causal, dtn) == "fa_fwd_pp16_kernel"
    # (head_dim 64 under a causal mask: dK/dV stays on the 32x32x16 kernel below 2^28 pairs per head - that shape is there for the forward)
    assert capi.kernel_name("dkdv", b, sq, sk, h, d, causal, dtn) == "fa_bwd_dkdv16_kernel" or (d == 64 and causal)
    q, k, v, do = _inputs(gpu, b, sq, sk, h, hk, dt, d)
    o, lse = _fwd(libs["product"], q, k, v, causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    grads = _bwd(libs["product"], q, k, v, o, lse, do, causal)
    for name in ("dma_late", "dma_sleepy"):
        for rep in range(2):
            o2, lse2 = _fwd(libs[name], q, k, v, causal)
            assert torch.equal(o, o2) and torch.equal(lse, lse2), (name, rep, (o.float() - o2.float()).abs().max().item())
        g2 = _bwd(libs[name], q, k, v, o, lse, do, causal)
        for a, c, t in zip(grads, g2, ("dq", "dk", "dv")):
            assert torch.equal(a, c), (name, t, (a.float() - c.float()).abs().max().item())


def test_late_issue_catches_the_documented_racy_form(gpu, libs):
    """the detector must fire on the bug it exists for: round 4's counted-wait form (K(u+2) left in flight across the barrier, the other group's half of
    the tile retired one barrier after the first read) is bit-identical to the product at normal timing - recorded, not asserted: it is a race - and
    wrong under late issue on every shape that reaches the steady loop"""
    seen_normal_equal = 0
    for _, b, sq, sk, h, hk, dt, causal in SHAPES[:3]:
        q, k, v, _ = _inputs(gpu, b, sq, sk, h, hk, dt)
        o, lse = _fwd(libs["product"], q, k, v, causal)
        o_r, _ = _fwd(libs["dma_racy"], q, k, v, causal)
        seen_normal_equal += int(torch.equal(o, o_r))
        o_rl, _ = _fwd(libs["dma_racy_late"], q, k, v, causal)
        diff = (o.float() - o_rl.float()).abs()
        assert not torch.equal(o, o_rl) and torch.nan_to_num(diff, nan=1.0).max().item() > 1e-2, "the late-issue build did not expose the racy form"
    print(f"racy form at normal timing: bit-identical to the product on {seen_normal_equal} of 3 shapes (the race is invisible to value tests)")
