"""ctypes/numpy front-end of the CPU oracle (oracle/attn_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Restates the reference algorithm (see the C file header for file:line citations); pinned
against the reference's own Python oracles by oracle/pin_oracle.py and tests/golden/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libattn_oracle.so")
_lib = None

ROUND_NONE, ROUND_FP16, ROUND_BF16 = 0, 1, 2


def build(force=False):
    src = os.path.join(_HERE, "attn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libattn_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.oracle_attn_fwd.argtypes = [fp, fp, fp, fp, fp, ip, ip] + [ctypes.c_int] * 8
        _lib.oracle_attn_fwd.restype = None
        _lib.oracle_attn_bwd.argtypes = [fp] * 9 + [ip, ip] + [ctypes.c_int] * 8
        _lib.oracle_attn_bwd.restype = None
        _lib.oracle_dot_do_o.argtypes = [fp, fp, fp, ip] + [ctypes.c_int] * 4
        _lib.oracle_dot_do_o.restype = None
        _lib.oracle_round_fp16.argtypes = [ctypes.c_float]
        _lib.oracle_round_fp16.restype = ctypes.c_float
        _lib.oracle_round_bf16.argtypes = [ctypes.c_float]
        _lib.oracle_round_bf16.restype = ctypes.c_float
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def num_threads():
    return int(lib().oracle_num_threads())


def attn_fwd(q, k, v, causal=False, round_mode=ROUND_FP16, cu_seqlens_q=None, cu_seqlens_k=None,
             max_seqlen_q=None, max_seqlen_k=None):
    """q (b,sq,h,d) / k,v (b,sk,hk,d) float arrays -> (o like q, lse (b,h,sq)) float32.
    varlen: q (total_q,h,d), k/v (total_k,hk,d), int32 cu_seqlens -> lse (b,h,max_seqlen_q) zero padded."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    if cu_seqlens_q is None:
        b, sq, h, d = q.shape
        sk, hk = k.shape[1], k.shape[2]
        cq = ck = None
    else:
        cq = np.ascontiguousarray(cu_seqlens_q, dtype=np.int32)
        ck = np.ascontiguousarray(cu_seqlens_k, dtype=np.int32)
        b = len(cq) - 1
        h, d = q.shape[1], q.shape[2]
        hk = k.shape[1]
        sq = int(max_seqlen_q if max_seqlen_q is not None else np.diff(cq).max(initial=0))
        sk = int(max_seqlen_k if max_seqlen_k is not None else np.diff(ck).max(initial=0))
    o = np.zeros_like(q)
    lse = np.zeros((b, h, sq), dtype=np.float32)
    lib().oracle_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), _ip(cq), _ip(ck), b, sq, sk, h, hk, d, int(causal), int(round_mode))
    return o, lse


def attn_bwd(q, k, v, o, lse, dout, causal=False, round_mode=ROUND_FP16, cu_seqlens_q=None, cu_seqlens_k=None,
             max_seqlen_q=None, max_seqlen_k=None):
    q, k, v, o, lse, dout = _f32(q), _f32(k), _f32(v), _f32(o), _f32(lse), _f32(dout)
    if cu_seqlens_q is None:
        b, sq, h, d = q.shape
        sk, hk = k.shape[1], k.shape[2]
        cq = ck = None
    else:
        cq = np.ascontiguousarray(cu_seqlens_q, dtype=np.int32)
        ck = np.ascontiguousarray(cu_seqlens_k, dtype=np.int32)
        b = len(cq) - 1
        h, d = q.shape[1], q.shape[2]
        hk = k.shape[1]
        sq = int(lse.shape[2])
        sk = int(max_seqlen_k if max_seqlen_k is not None else np.diff(ck).max(initial=0))
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    lib().oracle_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(dout), _p(dq), _p(dk), _p(dv), _ip(cq), _ip(ck),
                          b, sq, sk, h, hk, d, int(causal), int(round_mode))
    return dq, dk, dv


def dot_do_o(o, dout, cu_seqlens_q=None, max_seqlen_q=None):
    o, dout = _f32(o), _f32(dout)
    if cu_seqlens_q is None:
        b, sq, h, d = o.shape
        cq = None
    else:
        cq = np.ascontiguousarray(cu_seqlens_q, dtype=np.int32)
        b, h, d = len(cq) - 1, o.shape[1], o.shape[2]
        sq = int(max_seqlen_q)
    dsum = np.zeros((b, h, sq), dtype=np.float32)
    lib().oracle_dot_do_o(_p(o), _p(dout), _p(dsum), _ip(cq), b, sq, h, d)
    return dsum


def round_lp(x, round_mode):
    """elementwise round-to-nearest-even to fp16/bf16, returned as float32 (numpy reference of the C helpers)"""
    x = np.asarray(x, dtype=np.float32)
    if round_mode == ROUND_FP16:
        with np.errstate(over="ignore"):
            return x.astype(np.float16).astype(np.float32)
    if round_mode == ROUND_BF16:
        u = x.view(np.uint32).astype(np.uint64)
        nan_inf = (u & 0x7F800000) == 0x7F800000
        r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
        r = np.where(nan_inf, x.view(np.uint32), r).astype(np.uint32)
        return r.view(np.float32)
    return x
