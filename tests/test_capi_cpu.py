"""CPU: the C-ABI library loads, exports every symbol the header declares, and its host-side
validation works without a GPU (no compute calls here)."""
import ctypes

import pytest
import torch

from flash_attn_turing import capi


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = capi.declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/flash_attn_gfx950.h but not exported"
    assert L.fa_abi_version() == 4
    assert b"gfx950" in L.fa_build_info()


def test_forward_kernel_policy_is_host_state():
    """fa_set_kernel_policy: returns the previous value, refuses unknown ones, and fa_fwd_kernel_name follows it (no GPU involved)"""
    assert capi.set_kernel_policy(capi.POLICY_MFMA32) == capi.POLICY_AUTO      # the library's initial policy
    assert capi.fwd_kernel_name(128) == "fa_fwd_pp_kernel"
    assert capi.lib().fa_set_kernel_policy(3) == -1 and capi.lib().fa_set_kernel_policy(-1) == -1
    assert capi.set_kernel_policy(capi.POLICY_AUTO) == capi.POLICY_MFMA32
    assert capi.fwd_kernel_name(128) == "fa_fwd_pp16_kernel" and capi.fwd_kernel_name(64) == "fa_fwd_pp16_kernel"      # (64: fp16, round 4)
    # fa_kernel_name: the default policy's choices at the BASELINE shapes and at small ones
    assert capi.kernel_name("fwd", 4, 16384, 16384, 32, 128, True) == "fa_fwd_pp16_kernel"
    assert capi.kernel_name("fwd", 4, 4096, 4096, 32, 128, False) == "fa_fwd_pp16_kernel"
    assert capi.kernel_name("fwd", 1, 512, 512, 4, 128, False) == "fa_fwd_pp_kernel"
    # head_dim 64: fp16 from 2^24 pairs per head (2^26 under a causal mask) on the 16x16x32 forward when the launch fills the chip, from 2k x 2k (1k x 1k causal) when it
    # leaves the second workgroup slot of the 32x32x16 kernel empty (round 6); bf16 never unless pinned
    assert capi.kernel_name("fwd", 4, 8192, 8192, 32, 64, False) == "fa_fwd_pp16_kernel"
    assert capi.kernel_name("fwd", 4, 4096, 4096, 32, 64, False) == "fa_fwd_pp16_kernel"
    assert capi.kernel_name("fwd", 4, 2048, 2048, 32, 64, False) == "fa_fwd_pp_kernel"
    assert capi.kernel_name("fwd", 1, 2048, 2048, 32, 64, False) == "fa_fwd_pp16_kernel"          # 256 workgroups
    assert capi.kernel_name("fwd", 4, 4096, 4096, 32, 64, True) == "fa_fwd_pp_kernel"
    assert capi.kernel_name("fwd", 1, 4096, 4096, 32, 64, True) == capi.kernel_name("fwd", 1, 1024, 1024, 32, 64, True) == "fa_fwd_pp16_kernel"
    assert capi.kernel_name("fwd", 4, 8192, 8192, 32, 64, True) == "fa_fwd_pp16_kernel"
    assert capi.kernel_name("fwd", 4, 8192, 8192, 32, 64, False, "bf16") == capi.kernel_name("fwd", 4, 16384, 16384, 32, 64, True, "bf16") == capi.kernel_name("fwd", 1, 2048, 2048, 32, 64, False, "bf16") == "fa_fwd_pp_kernel"
    assert capi.kernel_name("fwd", 4, 8192, 8192, 32, 128, False, "bf16") == "fa_fwd_pp16_kernel"
    assert capi.lib().fa_kernel_name(0, 4, 8192, 8192, 32, 64, 0) == capi.lib().fa_kernel_name_dtype(0, 0, 4, 8192, 8192, 32, 64, 0)      # fa_kernel_name = fp16
    assert capi.lib().fa_kernel_name_dtype(0, 7, 4, 8192, 8192, 32, 64, 0) == b""
    assert capi.kernel_name("dq", 4, 8192, 8192, 32, 128, False) == "fa_bwd_dq16_kernel"
    # round 6: forward and dQ by how far the LAUNCH fills the chip (256 CUs assumed without a device): >= 1 workgroup of 256 query rows per CU, >= 2 under a causal mask
    # (or 1 from 8k x 8k), and >= 2^20 pairs per head for the forward and causal dQ
    assert capi.kernel_name("dq", 4, 8192, 8192, 32, 128, True) == "fa_bwd_dq16_kernel"           # 4096 workgroups (rounds 4-5: causal dQ from 2^28 pairs per head only)
    assert capi.kernel_name("dq", 4, 16384, 16384, 32, 128, True) == "fa_bwd_dq16_kernel"
    assert capi.kernel_name("dq", 1, 4096, 4096, 8, 128, False) == "fa_bwd_dq_kernel"             # 128 workgroups: the chip is half empty, the 32x32x16 kernel is 5 % ahead
    assert capi.kernel_name("dq", 1, 8192, 8192, 8, 128, False) == "fa_bwd_dq16_kernel"           # 256
    assert capi.kernel_name("dq", 1, 8192, 8192, 8, 128, True) == "fa_bwd_dq_kernel"              # 256 under a mask: not yet
    assert capi.kernel_name("dq", 1, 4096, 4096, 32, 128, True) == "fa_bwd_dq16_kernel"           # 512
    assert capi.kernel_name("dq", 64, 512, 512, 32, 128, True) == "fa_bwd_dq_kernel"              # many workgroups, but 2^18 pairs per head
    assert capi.kernel_name("fwd", 4, 4096, 4096, 32, 128, True) == "fa_fwd_pp16_kernel"
    assert capi.kernel_name("fwd", 4, 2048, 2048, 32, 128, True) == "fa_fwd_pp16_kernel"          # 1024 workgroups (round 5: from 2^23 pairs per head)
    assert capi.kernel_name("fwd", 4, 1024, 1024, 32, 128, False) == "fa_fwd_pp16_kernel"         # 512
    assert capi.kernel_name("fwd", 1, 1024, 1024, 32, 128, False) == "fa_fwd_pp_kernel"           # 128 workgroups
    assert capi.kernel_name("fwd", 1, 2048, 2048, 32, 128, False) == "fa_fwd_pp16_kernel"         # 256
    assert capi.kernel_name("fwd", 1, 2048, 2048, 32, 128, True) == "fa_fwd_pp_kernel"            # 256 under a mask
    assert capi.kernel_name("fwd", 1, 8192, 8192, 8, 128, True) == "fa_fwd_pp16_kernel"           # 256 under a mask, 2^26 pairs
    assert capi.kernel_name("fwd", 64, 512, 512, 32, 128, False) == "fa_fwd_pp_kernel"            # 2^18 pairs per head
    assert capi.kernel_name("dkdv", 4, 8192, 8192, 32, 128, False) == "fa_bwd_dkdv16_kernel"
    assert capi.kernel_name("dkdv", 4, 2048, 2048, 32, 128, False) == "fa_bwd_dkdv16_kernel"
    assert capi.kernel_name("dkdv", 4, 512, 512, 32, 128, True) == "fa_bwd_dkdv16_kernel"        # round 6: 512 x 512 when the launch has two workgroups per CU (here 512)
    assert capi.kernel_name("dkdv", 1, 512, 512, 32, 128, True) == capi.kernel_name("dkdv", 64, 256, 256, 32, 128, False) == "fa_bwd_dkdv_kernel"
    # per launch (round 6) - unless the caller states the whole problem's batch x heads: a (batch, head) shard then gets the kernel of the whole problem
    assert capi.kernel_name("fwd", 1, 16384, 16384, 1, 128, True) == "fa_fwd_pp_kernel" and capi.kernel_name("fwd", 64, 16384, 16384, 64, 128, True) == "fa_fwd_pp16_kernel"
    assert capi.set_policy_problem_heads(64 * 64) == 0
    assert capi.kernel_name("fwd", 1, 16384, 16384, 1, 128, True) == "fa_fwd_pp16_kernel" and capi.kernel_name("dq", 1, 16384, 16384, 1, 128, True) == "fa_bwd_dq16_kernel"
    assert capi.set_policy_problem_heads(0) == 64 * 64 and capi.lib().fa_set_policy_problem_heads(-5) == -1 and capi.set_policy_problem_heads(0) == 0
    assert capi.kernel_name("fwd", 1, 16384, 16384, 1, 128, True) == "fa_fwd_pp_kernel"
    assert capi.kernel_name("dkdv", 1, 8192, 8192, 2, 128, True) == "fa_bwd_dkdv16_kernel"       # dK/dV: by the pairs per head whatever the launch
    # head_dim 64 backward (round 5): dQ 16x16x32 without a mask from 2^18 pairs per head (round 6) and under one from 2^26; dK/dV from 2^24 (2^28 causal);
    # never when a causal problem has fewer keys than queries (dead row blocks); both dtypes
    assert capi.kernel_name("dkdv", 4, 8192, 8192, 32, 64, False) == capi.kernel_name("dkdv", 1, 4096, 4096, 1, 64, False, "bf16") == "fa_bwd_dkdv16_kernel"
    assert capi.kernel_name("dkdv", 4, 2048, 2048, 32, 64, False) == capi.kernel_name("dkdv", 4, 8192, 8192, 32, 64, True) == "fa_bwd_dkdv_kernel"
    assert capi.kernel_name("dkdv", 4, 16384, 16384, 32, 64, True) == "fa_bwd_dkdv16_kernel"
    assert capi.kernel_name("dq", 4, 512, 512, 32, 64, False) == capi.kernel_name("dq", 4, 8192, 8192, 32, 64, True) == "fa_bwd_dq16_kernel"
    assert capi.kernel_name("dq", 4, 256, 256, 32, 64, False) == capi.kernel_name("dq", 1, 1, 1, 1, 64, False) == "fa_bwd_dq_kernel"      # round 6: non-causal dQ from 2^18 pairs per head (was: always)
    assert capi.kernel_name("dq", 4, 4096, 4096, 32, 64, True) == capi.kernel_name("dq", 4, 16384, 8192, 32, 64, True) == "fa_bwd_dq_kernel"
    # round 6: launches that leave the 32x32x16 kernels' second workgroup slot empty go to the 16x16x32 set at every length; many short sequences keep the 32x32x16 dQ
    assert capi.kernel_name("dq", 1, 4096, 4096, 32, 64, True) == capi.kernel_name("dq", 1, 512, 512, 8, 64, True) == "fa_bwd_dq16_kernel"
    assert capi.kernel_name("dkdv", 1, 1024, 1024, 32, 64, False) == capi.kernel_name("dkdv", 1, 4096, 4096, 32, 64, True) == capi.kernel_name("dkdv", 1, 2048, 2048, 8, 64, True) == "fa_bwd_dkdv16_kernel"
    assert capi.kernel_name("dkdv", 1, 8192, 8192, 32, 64, True) == capi.kernel_name("dkdv", 4, 1024, 1024, 32, 64, True) == "fa_bwd_dkdv_kernel"
    assert capi.kernel_name("dq", 16, 512, 512, 32, 64, False) == "fa_bwd_dq_kernel" and capi.kernel_name("dq", 16, 2048, 2048, 32, 64, False) == "fa_bwd_dq16_kernel"
    assert capi.lib().fa_kernel_name(9, 1, 1, 1, 1, 128, 0) == b""


def test_package_level_policy_helpers():
    import flash_attn_turing as F

    assert F.set_kernel_policy("mfma16") == "auto"
    assert F.kernel_name("fwd", 1, 128, 128, 1, 128, False) == "fa_fwd_pp16_kernel"
    assert F.kernel_name("fwd", 1, 128, 128, 1, 64, False, torch.bfloat16) == "fa_fwd_pp16_kernel"      # pinned: both dtypes
    assert F.set_kernel_policy("auto") == "mfma16"
    assert F.kernel_name("fwd", 1, 8192, 8192, 1, 64, False, torch.bfloat16) == "fa_fwd_pp_kernel"
    assert F.kernel_name("fwd", 1, 8192, 8192, 1, 64, False, torch.float16) == F.kernel_name("fwd", 1, 8192, 8192, 1, 64, False) == "fa_fwd_pp16_kernel"
    with pytest.raises(ValueError):
        F.kernel_name("fwd", 1, 8192, 8192, 1, 64, False, torch.float32)
    assert F.kernel_name("fwd", 1, 128, 128, 1, 128, False) == "fa_fwd_pp_kernel"
    with pytest.raises(ValueError):
        F.set_kernel_policy("fastest")


def test_flops_and_bytes_match_survey_figures():
    L = capi.lib()
    # C2 1.0995e12, C3 non-causal 1.7592e13
    assert L.fa_fwd_flops(4, 4096, 4096, 32, 128, 0) == pytest.approx(1.0995e12, rel=1e-4)
    assert L.fa_fwd_flops(4, 16384, 16384, 32, 128, 0) == pytest.approx(1.7592e13, rel=1e-4)
    # causal counts visible pairs exactly: s(s+1)/2
    assert L.fa_fwd_flops(1, 4, 4, 1, 128, 1) == 4 * 10 * 128
    assert L.fa_fwd_flops(1, 4, 2, 1, 64, 1) == 4 * (0 + 0 + 1 + 2) * 64      # bottom-right aligned, two dead rows
    assert L.fa_fwd_flops(1, 2, 4, 1, 64, 1) == 4 * (3 + 4) * 64
    assert L.fa_fwd_bytes(4, 4096, 4096, 32, 32, 128) == pytest.approx(514 * 2**20, rel=1e-3)


def test_host_validation_error_codes_without_gpu():
    L = capi.lib()
    p = capi.FwdParams()
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d, p.dtype = 1, 8, 8, 3, 2, 128, 0
    assert L.fa_run_mha_fwd(ctypes.byref(p), None) == capi.FA_ERR_BAD_GQA
    assert "divisible" in capi.last_error()
    p.h = 2
    p.d = 96
    assert L.fa_run_mha_fwd(ctypes.byref(p), None) == capi.FA_ERR_BAD_HEADDIM
    p.d = 128
    p.dtype = 7
    assert L.fa_run_mha_fwd(ctypes.byref(p), None) == capi.FA_ERR_BAD_DTYPE
    p.dtype = 0
    assert L.fa_run_mha_fwd(ctypes.byref(p), None) == capi.FA_ERR_NULL_POINTER      # lse NULL
    assert L.fa_run_mha_fwd(None, None) == capi.FA_ERR_NULL_POINTER
    b = capi.BwdParams()
    b.b, b.seqlen_q, b.seqlen_k, b.h, b.h_k, b.d, b.dtype = 1, 8, 8, 2, 2, 64, 1
    assert L.fa_run_mha_bwd(ctypes.byref(b), None) == capi.FA_ERR_NULL_POINTER
    # misaligned / non-dense strides are rejected before any launch
    buf = (ctypes.c_char * 4096)()
    addr = ctypes.addressof(buf)
    addr += (-addr) % 16
    p.q = p.k = p.v = p.o = p.lse = addr
    p.q_stride = p.k_stride = p.v_stride = p.o_stride = capi.Strides(8 * 2 * 128, 2 * 128 + 4, 128)
    assert L.fa_run_mha_fwd(ctypes.byref(p), None) == capi.FA_ERR_BAD_STRIDE
    # ABI 2: the optional packed-token totals must be >= 0 (0 = unknown); rejected before any launch
    p.q_stride = p.k_stride = p.v_stride = p.o_stride = capi.Strides(8 * 2 * 128, 2 * 128, 128)
    p.total_q = -1
    assert L.fa_run_mha_fwd(ctypes.byref(p), None) == capi.FA_ERR_BAD_SHAPE
    assert "total_q" in capi.last_error()


def test_struct_layout_matches_header():
    # ABI 4 header {struct_size, magic} + 7 pointers + 8 int32 + 4 x 3 int64 + 2 int64 (total_q, total_k);
    # header + 12 pointers + 8 int32 + 8 x 3 int64 + 2 int64 + workspace pointer + workspace_bytes
    assert ctypes.sizeof(capi.FwdParams) == 8 + 7 * 8 + 8 * 4 + 4 * 24 + 16
    assert ctypes.sizeof(capi.BwdParams) == 8 + 12 * 8 + 8 * 4 + 8 * 24 + 16 + 16
    assert capi.FwdParams.struct_size.offset == 0 and capi.FwdParams.magic.offset == 4 and capi.FwdParams.q.offset == 8
    # the optional fields sit at the very end, in the order they were appended
    assert capi.FwdParams.total_q.offset == 8 + 7 * 8 + 8 * 4 + 4 * 24 and capi.BwdParams.total_q.offset == 8 + 12 * 8 + 8 * 4 + 8 * 24
    assert capi.BwdParams.workspace.offset == capi.BwdParams.total_k.offset + 8
    assert capi.lib().fa_abi_version() == 4
    # the header's constant and the binding's agree
    import re
    with open(capi.HEADER_PATH) as f:
        assert int(re.search(r"#define FA_PARAMS_MAGIC (0x[0-9A-Fa-f]+)u", f.read()).group(1), 16) == capi.FA_PARAMS_MAGIC


def test_abi_header_guards_struct_size():
    """ABI 4: a struct without the {struct_size, magic} header (what an ABI 1-3 caller would pass: its q pointer sits there), one
    shorter than the mandatory part, or one longer than the library knows is FA_ERR_BAD_ABI, before anything is read past its end;
    a struct that stops before the appended optional fields is accepted and those fields default to "not given"."""
    L = capi.lib()
    p = _bwd_params_host_only(4, 8192, 8192, 32, 1, 128, True)
    full = capi.bwd_workspace_bytes(p)
    assert full > 0
    # (1) an ABI 3 caller: no header, first 8 bytes = a device pointer
    p.struct_size, p.magic = 0x7F3A1000, 0x00007F12
    assert L.fa_bwd_workspace_bytes(ctypes.byref(p)) == capi.FA_ERR_BAD_ABI and "ABI < 4" in capi.last_error()
    assert L.fa_run_mha_bwd(ctypes.byref(p), None) == capi.FA_ERR_BAD_ABI
    # (2) sizes outside [mandatory part, sizeof]
    p.magic = capi.FA_PARAMS_MAGIC
    for bad in (0, capi.BwdParams.total_q.offset - 8, ctypes.sizeof(capi.BwdParams) + 8):
        p.struct_size = bad
        assert L.fa_bwd_workspace_bytes(ctypes.byref(p)) == capi.FA_ERR_BAD_ABI, bad
    # (3) a caller whose struct ends before `workspace` (an older ABI >= 4 header): accepted; the fields it does not have are never
    # read - poison them to prove it
    p.struct_size = capi.BwdParams.workspace.offset
    p.workspace, p.workspace_bytes = 0xDEAD0001, -5
    assert L.fa_bwd_workspace_bytes(ctypes.byref(p)) == full
    # ... and one that ends before total_q / total_k: a packed batch then has no totals -> no split
    v = _bwd_params_host_only(2, 1024, 1024, 8, 1, 128, True, total_k=1500, varlen=True)
    assert capi.bwd_workspace_bytes(v) > 0
    v.struct_size = capi.BwdParams.total_q.offset
    assert capi.bwd_workspace_bytes(v) == 0
    f = capi.FwdParams()
    f.b, f.seqlen_q, f.seqlen_k, f.h, f.h_k, f.d, f.dtype = 1, 8, 8, 3, 2, 128, 0
    f.magic = 0
    assert L.fa_run_mha_fwd(ctypes.byref(f), None) == capi.FA_ERR_BAD_ABI


def _bwd_params_host_only(b, sq, sk, h, hk, d, causal, total_k=0, varlen=False):
    """fa_bwd_params over dummy 16-byte aligned addresses: enough for the host-side arithmetic entry points (no launch)"""
    buf = (ctypes.c_char * 64)()
    addr = ctypes.addressof(buf)
    addr += (-addr) % 16
    p = capi.BwdParams()
    for f in ("q", "k", "v", "o", "dout", "lse", "dq", "dk", "dv", "dsoftmax_sum"):
        setattr(p, f, addr)
    if varlen:
        p.cu_seqlens_q = p.cu_seqlens_k = addr
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d, p.dtype, p.is_causal = b, sq, sk, h, hk, d, 1, int(causal)
    qs, ks = capi.Strides(sq * h * d, h * d, d), capi.Strides(sk * hk * d, hk * d, d)
    p.q_stride = p.o_stride = p.do_stride = p.dq_stride = qs
    p.k_stride = p.v_stride = p.dk_stride = p.dv_stride = ks
    p.total_k = total_k
    p._keep = buf
    return p


def test_dkdv_workspace_rule_host_arithmetic():
    """fa_bwd_workspace_bytes: 0 for MHA and for grids that already fill the chip; planes of 2 x n_split x rows x h_k x d fp32 otherwise;
    split doubles while the group divides evenly and the grid is below 4 workgroups per CU (causal) / 1 per CU (no mask)"""
    W = lambda *a, **k: capi.bwd_workspace_bytes(_bwd_params_host_only(*a, **k))
    plane = lambda b, sk, hk, d: 2 * b * sk * hk * d * 4
    assert W(4, 8192, 8192, 32, 32, 128, True) == 0                              # MHA: never
    assert W(4, 8192, 8192, 32, 1, 128, True) == 4 * plane(4, 8192, 1, 128)      # MQA causal 8k: 256 workgroups -> x4 = 1024
    assert W(4, 8192, 8192, 32, 1, 128, False) == 0                              # same grid, no mask: one workgroup per CU already
    assert W(1, 2048, 2048, 32, 1, 128, False) == 16 * plane(1, 2048, 1, 128)    # 16 workgroups -> x16 = 256
    assert W(4, 4096, 4096, 32, 8, 128, True) == 0                               # GQA 32/8 at 4k: 1024 workgroups
    assert W(2, 1024, 1024, 6, 1, 64, True) == 2 * plane(2, 1024, 1, 64)         # group of 6: 2 divides, 4 does not
    assert W(2, 1024, 1024, 8, 1, 128, True, varlen=True) == 0                   # packed tensors without total_k: no split
    assert W(2, 1024, 1024, 8, 1, 128, True, total_k=1500, varlen=True) == 8 * 2 * 1500 * 128 * 4
    # uniform-length packed batches take the PLAIN grid (varlen_slot_count() == 0): they are sized like the dense equivalent, not
    # split to the hilt (round-2 advisor finding: b8 x 8192 GQA 32/8 asked for 2 GiB, MQA h64 b16 for 8 GiB)
    assert W(8, 8192, 8192, 32, 8, 128, False, total_k=8 * 8192, varlen=True) == W(8, 8192, 8192, 32, 8, 128, False) == 0
    assert W(16, 8192, 8192, 64, 1, 128, True, total_k=16 * 8192, varlen=True) == W(16, 8192, 8192, 64, 1, 128, True) == 0
    assert W(4, 8192, 8192, 32, 1, 128, True, total_k=4 * 8192, varlen=True) == W(4, 8192, 8192, 32, 1, 128, True)
    # more sequences than the compact grid handles (kVarlenMaxBatch = 512): plain grid as well
    assert W(600, 256, 256, 8, 1, 128, True, total_k=600 * 256, varlen=True) == W(600, 256, 256, 8, 1, 128, True) == 0
    # the request is bounded by construction: it is only made while the grid is below 4 x 256 workgroups of 128 keys
    for args in ((4, 8192, 8192, 32, 1, 128, True), (1, 2048, 2048, 32, 1, 128, False), (2, 1024, 1024, 6, 1, 64, True)):
        assert W(*args) <= 2 * 2 * (4 * 256) * 128 * 128 * 4
    # the `workspace` fields of the argument are ignored here: a stale / unaligned pointer or a negative size left in a reused struct
    # does not turn a size query into an error (they ARE validated by the launches that use them)
    p = _bwd_params_host_only(1, 128, 128, 8, 1, 128, True)
    want = capi.bwd_workspace_bytes(p)
    p.workspace, p.workspace_bytes = 0x1003, -1
    assert capi.lib().fa_bwd_workspace_bytes(ctypes.byref(p)) == want
    assert capi.lib().fa_bwd_dkdv(ctypes.byref(p), None) == capi.FA_ERR_BAD_SHAPE
    p.workspace_bytes = 64
    assert capi.lib().fa_bwd_dkdv(ctypes.byref(p), None) == capi.FA_ERR_BAD_STRIDE


def test_header_is_valid_c_and_links_from_plain_c(tmp_path):
    """include/flash_attn_gfx950.h is a C header: a C11 translation unit (gcc, no C++) includes it, fills the structs with FA_PARAMS_INIT,
    links against the library and gets the host-side validation codes back - what a cgo / JNI / FFI binding of the reference would do."""
    import os
    import subprocess

    src = tmp_path / "use_header.c"
    src.write_text(r"""
#include <stdio.h>
#include "flash_attn_gfx950.h"
int main(void) {
    fa_fwd_params f; fa_bwd_params b;
    FA_PARAMS_INIT(f); FA_PARAMS_INIT(b);
    if (f.struct_size != sizeof(f) || b.magic != FA_PARAMS_MAGIC) return 10;
    f.b = 1; f.seqlen_q = 8; f.seqlen_k = 8; f.h = 3; f.h_k = 2; f.d = 128; f.dtype = FA_FP16;
    if (fa_run_mha_fwd(&f, NULL) != FA_ERR_BAD_GQA) return 11;
    f.magic = 0;
    if (fa_run_mha_fwd(&f, NULL) != FA_ERR_BAD_ABI) return 12;
    if (fa_abi_version() != FA_ABI_VERSION) return 13;
    printf("%s\n", fa_build_info());
    return 0;
}
""")
    exe = tmp_path / "use_header"
    libdir = os.path.dirname(capi.LIBRARY_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.dirname(capi.HEADER_PATH), str(src), "-o", str(exe),
                           "-L", libdir, "-l:libflash_attn_gfx950.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert "abi=4" in out.stdout and "src=" in out.stdout
