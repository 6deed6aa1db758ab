"""CPU: the C-ABI library loads, exports every symbol the header declares, and its host-side
validation works without a GPU (no compute calls here)."""
import ctypes

import pytest

from flash_attn_turing import capi


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = capi.declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/flash_attn_gfx950.h but not exported"
    assert L.fa_abi_version() == 2
    assert b"gfx950" in L.fa_build_info()


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
    # 7 pointers + 8 int32 + 4 x 3 int64 + 2 int64 (ABI 2: total_q, total_k) ; 12 pointers + 8 int32 + 8 x 3 int64 + 2 int64
    assert ctypes.sizeof(capi.FwdParams) == 7 * 8 + 8 * 4 + 4 * 24 + 16
    assert ctypes.sizeof(capi.BwdParams) == 12 * 8 + 8 * 4 + 8 * 24 + 16
    # the ABI 1 prefix is unchanged: the appended fields sit at the very end
    assert capi.FwdParams.total_q.offset == 7 * 8 + 8 * 4 + 4 * 24 and capi.BwdParams.total_q.offset == 12 * 8 + 8 * 4 + 8 * 24
