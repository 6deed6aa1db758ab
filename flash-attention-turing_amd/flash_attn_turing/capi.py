"""ctypes binding of the C ABI in include/flash_attn_gfx950.h (libflash_attn_gfx950.so).

This is the binding a non-PyTorch host (or the reference's maintainers)
would write: plain device pointers and sizes, a stream handle, integer status codes.  The
PyTorch host module (`_C`) is built on the same entry points in C++; this module exists so
tests and bench.py can drive and time the C ABI directly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))


def _first_existing(*candidates):
    return next((c for c in candidates if os.path.exists(c)), candidates[-1])


# installed package (setup.py ships library and header inside it) first, then the in-tree build
LIBRARY_PATH = _first_existing(os.path.join(_HERE, "libflash_attn_gfx950.so"), os.path.join(os.path.dirname(_HERE), "csrc", "libflash_attn_gfx950.so"))
# development aid (tools/ only: A/B and error-distribution scripts drive a variant build through this binding); the host module `_C` links the product library
if os.environ.get("FA_GFX950_LIBRARY"):
    LIBRARY_PATH = os.path.abspath(os.environ["FA_GFX950_LIBRARY"])
HEADER_PATH = _first_existing(os.path.join(_HERE, "include", "flash_attn_gfx950.h"),
                              os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "flash_attn_gfx950.h"))

FA_FP16, FA_BF16 = 0, 1
FA_OK = 0
FA_ERR_NULL_POINTER, FA_ERR_BAD_SHAPE, FA_ERR_BAD_GQA, FA_ERR_BAD_HEADDIM, FA_ERR_BAD_DTYPE, FA_ERR_BAD_STRIDE = -1, -2, -3, -4, -5, -6
FA_ERR_BAD_ABI = -8
FA_PARAMS_MAGIC = 0xFA950A71      # include/flash_attn_gfx950.h

_vp, _i32, _fp = ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p


class Strides(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int64), ("row", ctypes.c_int64), ("head", ctypes.c_int64)]


class _Params(ctypes.Structure):
    """ABI 4: every params struct starts with {struct_size, magic}; the constructor fills them like FA_PARAMS_INIT"""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size, self.magic = ctypes.sizeof(type(self)), FA_PARAMS_MAGIC


class FwdParams(_Params):
    _fields_ = [("struct_size", ctypes.c_uint32), ("magic", ctypes.c_uint32), ("q", _vp), ("k", _vp), ("v", _vp), ("o", _vp), ("lse", _vp),
                ("cu_seqlens_q", _vp), ("cu_seqlens_k", _vp),
                ("b", _i32), ("seqlen_q", _i32), ("seqlen_k", _i32), ("h", _i32), ("h_k", _i32), ("d", _i32),
                ("dtype", _i32), ("is_causal", _i32),
                ("q_stride", Strides), ("k_stride", Strides), ("v_stride", Strides), ("o_stride", Strides),
                ("total_q", ctypes.c_int64), ("total_k", ctypes.c_int64)]      # ABI 2, optional (0 = unknown)


class BwdParams(_Params):
    _fields_ = [("struct_size", ctypes.c_uint32), ("magic", ctypes.c_uint32), ("q", _vp), ("k", _vp), ("v", _vp), ("o", _vp), ("dout", _vp), ("lse", _vp),
                ("dq", _vp), ("dk", _vp), ("dv", _vp), ("dsoftmax_sum", _vp),
                ("cu_seqlens_q", _vp), ("cu_seqlens_k", _vp),
                ("b", _i32), ("seqlen_q", _i32), ("seqlen_k", _i32), ("h", _i32), ("h_k", _i32), ("d", _i32),
                ("dtype", _i32), ("is_causal", _i32),
                ("q_stride", Strides), ("k_stride", Strides), ("v_stride", Strides), ("o_stride", Strides),
                ("do_stride", Strides), ("dq_stride", Strides), ("dk_stride", Strides), ("dv_stride", Strides),
                ("total_q", ctypes.c_int64), ("total_k", ctypes.c_int64),
                ("workspace", _vp), ("workspace_bytes", ctypes.c_int64)]


_lib = None


def declared_functions():
    """names of every function declared in the public header"""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBRARY_PATH):
            raise ImportError(f"{LIBRARY_PATH} missing: run `python flash-attention-turing_amd/build.py`")
        L = ctypes.CDLL(LIBRARY_PATH)
        # a stale build of the same ABI number can lack the (additive) exports this binding needs: say so instead of failing on first use
        missing = [n for n in declared_functions() if not hasattr(L, n)]
        if missing:
            raise ImportError(f"{LIBRARY_PATH} lacks {missing}: stale build, run `python flash-attention-turing_amd/build.py --force`")
        L.fa_abi_version.restype = ctypes.c_int
        L.fa_last_error.restype = ctypes.c_char_p
        L.fa_build_info.restype = ctypes.c_char_p
        L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(FwdParams), _vp]
        L.fa_run_mha_bwd.argtypes = [ctypes.POINTER(BwdParams), _vp]
        L.fa_bwd_dot_do_o.argtypes = [ctypes.POINTER(BwdParams), _vp]
        L.fa_bwd_dq.argtypes = [ctypes.POINTER(BwdParams), _vp]
        L.fa_bwd_dkdv.argtypes = [ctypes.POINTER(BwdParams), _vp]
        L.fa_bwd_workspace_bytes.argtypes = [ctypes.POINTER(BwdParams)]
        L.fa_bwd_workspace_bytes.restype = ctypes.c_int64
        L.fa_mha_fwd.argtypes = [_vp] * 5 + [_i32] * 8 + [_vp]
        L.fa_mha_bwd.argtypes = [_vp] * 10 + [_i32] * 8 + [_vp]
        L.fa_mha_varlen_fwd.argtypes = [_vp] * 7 + [_i32] * 8 + [_vp]
        L.fa_mha_varlen_bwd.argtypes = [_vp] * 12 + [_i32] * 8 + [_vp]
        for n in ("fa_run_mha_fwd", "fa_run_mha_bwd", "fa_bwd_dot_do_o", "fa_bwd_dq", "fa_bwd_dkdv", "fa_mha_fwd", "fa_mha_bwd",
                  "fa_mha_varlen_fwd", "fa_mha_varlen_bwd"):
            getattr(L, n).restype = ctypes.c_int
        L.fa_fwd_flops.argtypes = [_i32] * 6
        L.fa_fwd_flops.restype = ctypes.c_double
        L.fa_fwd_bytes.argtypes = [_i32] * 6
        L.fa_fwd_bytes.restype = ctypes.c_double
        L.fa_device_clock_khz.argtypes = [_i32]
        L.fa_device_clock_khz.restype = ctypes.c_int
        L.fa_fwd_kernel_name.argtypes = [_i32]
        L.fa_fwd_kernel_name.restype = ctypes.c_char_p
        L.fa_kernel_name.argtypes = [_i32] * 7
        L.fa_kernel_name.restype = ctypes.c_char_p
        L.fa_kernel_name_dtype.argtypes = [_i32] * 8
        L.fa_kernel_name_dtype.restype = ctypes.c_char_p
        L.fa_set_kernel_policy.argtypes = [_i32]
        L.fa_set_kernel_policy.restype = _i32
        L.fa_set_policy_problem_heads.argtypes = [ctypes.c_int64]
        L.fa_set_policy_problem_heads.restype = ctypes.c_int64
        _lib = L
    return _lib


def device_clock_khz(device_index=0) -> int:
    return int(lib().fa_device_clock_khz(int(device_index)))


def fwd_kernel_name(d) -> str:
    return lib().fa_fwd_kernel_name(int(d)).decode()


STAGES = {"fwd": 0, "dq": 1, "dkdv": 2}


DTYPES = {"fp16": 0, "bf16": 1}


def kernel_name(stage, b, seqlen_q, seqlen_k, h, d, causal, dtype="fp16") -> str:
    """fa_kernel_name_dtype: the kernel a launch of this shape goes to under the current policy (stage: "fwd", "dq", "dkdv"; dtype "fp16" / "bf16")"""
    return lib().fa_kernel_name_dtype(STAGES[stage], DTYPES[dtype], int(b), int(seqlen_q), int(seqlen_k), int(h), int(d), int(bool(causal))).decode()


POLICY_MFMA32, POLICY_MFMA16, POLICY_AUTO = 0, 1, 2


def set_kernel_policy(policy) -> int:
    """fa_set_kernel_policy: which head_dim-128 kernel set (32x32x16 / 16x16x32 MFMA tiles) serves a launch; returns the previous policy"""
    prev = lib().fa_set_kernel_policy(int(policy))
    if prev < 0:
        raise ValueError(f"unknown kernel policy {policy}")
    return prev


def set_policy_problem_heads(batch_times_heads) -> int:
    """fa_set_policy_problem_heads: the batch x heads FA_POLICY_AUTO sizes the following launches with (0 = each launch's own); returns the previous value"""
    prev = lib().fa_set_policy_problem_heads(int(batch_times_heads))
    if prev < 0:
        raise ValueError(f"batch x heads must be >= 0, got {batch_times_heads}")
    return prev


def last_error() -> str:
    return lib().fa_last_error().decode()


def check(rc: int):
    if rc != FA_OK:
        raise RuntimeError(f"flash_attn_gfx950: {last_error()} [code {rc}]")


def dtype_code(torch_dtype) -> int:
    import torch

    return {torch.float16: FA_FP16, torch.bfloat16: FA_BF16}[torch_dtype]


def mha_fwd(q, k, v, o, lse, causal, stream=None):
    """Contiguous torch tensors in, kernels enqueued on `stream` (default: torch's current stream)."""
    import torch

    b, sq, h, d = q.shape
    sk, hk = k.shape[1], k.shape[2]
    s = torch.cuda.current_stream(q.device).cuda_stream if stream is None else stream
    check(lib().fa_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(),
                           b, sq, sk, h, hk, d, dtype_code(q.dtype), int(causal), s))


def mha_bwd(q, k, v, o, lse, dout, dq, dk, dv, dsum, causal, stream=None):
    import torch

    b, sq, h, d = q.shape
    sk, hk = k.shape[1], k.shape[2]
    s = torch.cuda.current_stream(q.device).cuda_stream if stream is None else stream
    check(lib().fa_mha_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), dout.data_ptr(),
                           dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dsum.data_ptr(),
                           b, sq, sk, h, hk, d, dtype_code(q.dtype), int(causal), s))


def bwd_params(q, k, v, o, lse, dout, dq, dk, dv, dsum, causal):
    """fa_bwd_params for contiguous (b, s, h, d) torch tensors (stage-level entry points: fa_bwd_dot_do_o / fa_bwd_dq / fa_bwd_dkdv)"""
    b, sq, h, d = q.shape
    sk, hk = k.shape[1], k.shape[2]
    p = BwdParams()
    p.q, p.k, p.v, p.o, p.dout, p.lse = (t.data_ptr() for t in (q, k, v, o, dout, lse))
    p.dq, p.dk, p.dv, p.dsoftmax_sum = (t.data_ptr() for t in (dq, dk, dv, dsum))
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d = b, sq, sk, h, hk, d
    p.dtype, p.is_causal = dtype_code(q.dtype), int(causal)
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", o), ("do_stride", dout),
                    ("dq_stride", dq), ("dk_stride", dk), ("dv_stride", dv)):
        setattr(p, name, Strides(t.stride(0), t.stride(1), t.stride(2)))
    return p


def bwd_workspace_bytes(params):
    """bytes of fp32 scratch the dK/dV launch of these params would use to split a KV head's query-head group (0: no split)"""
    n = lib().fa_bwd_workspace_bytes(ctypes.byref(params))
    if n < 0:
        check(int(n))
    return int(n)


def attach_workspace(params, like):
    """allocate the scratch fa_bwd_workspace_bytes asks for (torch, on `like`'s device) and hang it on params; returns the tensor
    (keep it alive until the launch has run) or None"""
    import torch

    n = bwd_workspace_bytes(params)
    if n == 0:
        return None
    ws = torch.empty(n // 4, device=like.device, dtype=torch.float32)
    params.workspace, params.workspace_bytes = ws.data_ptr(), n
    return ws


def run_bwd(params, stream=None):
    """the whole backward from a parameter struct, e.g. one with a workspace attached: dQ (which also computes D = rowsum(dO * O)
    into dsoftmax_sum), then dK/dV (+ the plane sum when the head group was split through the workspace)"""
    import torch

    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    check(lib().fa_run_mha_bwd(ctypes.byref(params), s))


def bwd_stage(name, params, stream=None):
    """run one backward launch: name in {'dot_do_o', 'dq', 'dkdv'}"""
    import torch

    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    check(getattr(lib(), "fa_bwd_" + name)(ctypes.byref(params), s))


def fwd_params(q, k, v, o, lse, causal):
    """fa_fwd_params for (b, s, h, d) torch tensors with arbitrary batch / row / head strides (e.g. shard views)"""
    b, sq, h, d = q.shape
    sk, hk = k.shape[1], k.shape[2]
    p = FwdParams()
    p.q, p.k, p.v, p.o, p.lse = (t.data_ptr() for t in (q, k, v, o, lse))
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d = b, sq, sk, h, hk, d
    p.dtype, p.is_causal = dtype_code(q.dtype), int(causal)
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v), ("o_stride", o)):
        setattr(p, name, Strides(t.stride(0), t.stride(1), t.stride(2)))
    return p


def run_fwd(params, stream=None):
    import torch

    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    check(lib().fa_run_mha_fwd(ctypes.byref(params), s))
