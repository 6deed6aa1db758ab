"""Autograd-aware wrappers over the raw fwd/bwd entry points.

The reference ships only the four raw functions; its README (README.md:28-48) documents
``flash_attn_func(q, k, v, batch_size, seq_len, num_heads, head_dim)`` from an older API.
``flash_attn_func`` here accepts both that legacy call (the four ints are validated against
the tensor shapes and otherwise ignored) and the modern ``flash_attn_func(q, k, v, causal=False)``.
"""
import torch

from . import _C


class FlashAttnFunc(torch.autograd.Function):
    """O = softmax(Q K^T / sqrt(d) + causal_mask) V on (batch, seqlen, heads, head_dim) tensors.

    The Python form of the autograd node, kept as the readable statement of what ``_C.attn_autograd`` does (flash_api.cpp:FlashAttnNode);
    ``flash_attn_func`` goes through the C++ node, which costs the host 42-52 us per forward + backward instead of ~85 (profiles/r4_host_overhead.log)."""

    @staticmethod
    def forward(ctx, q, k, v, causal):
        out, lse = _C.fwd(q, k, v, bool(causal))
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.causal = bool(causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = _C.bwd(q, k, v, out, lse, dout, ctx.causal)      # strided dout is fine: the kernels take real strides
        return dq, dk, dv, None


class FlashAttnVarlenFunc(torch.autograd.Function):
    """Packed variable-length variant: q (total_q, h, d), k/v (total_k, h_k, d), int32 cu_seqlens."""

    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal):
        out, lse = _C.varlen_fwd(q, k, v, cu_seqlens_q, cu_seqlens_k, int(max_seqlen_q), int(max_seqlen_k), bool(causal))
        ctx.save_for_backward(q, k, v, out, lse, cu_seqlens_q, cu_seqlens_k)
        ctx.meta = (int(max_seqlen_q), int(max_seqlen_k), bool(causal))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, cu_q, cu_k = ctx.saved_tensors
        max_q, max_k, causal = ctx.meta
        dq, dk, dv = _C.varlen_bwd(q, k, v, out, lse, dout, cu_q, cu_k, max_q, max_k, causal)
        return dq, dk, dv, None, None, None, None, None


def flash_attn_func(q, k, v, *legacy_dims, causal=False, return_lse=False):
    """Fused scaled-dot-product attention (scale fixed at 1/sqrt(head_dim), like the reference).

    q: (batch, seqlen_q, nheads, d); k, v: (batch, seqlen_k, nheads_k, d); fp16 or bf16 on a
    ROCm device.  ``causal`` uses bottom-right alignment (reference mask.h:172).  The legacy
    README form ``flash_attn_func(q, k, v, batch_size, seq_len, num_heads, head_dim)`` is accepted.
    """
    if legacy_dims:
        if len(legacy_dims) == 1 and isinstance(legacy_dims[0], bool):
            causal = legacy_dims[0]
        elif len(legacy_dims) == 4:
            b, s, h, d = (int(x) for x in legacy_dims)
            if (b, s, h, d) != (q.shape[0], q.shape[1], q.shape[2], q.shape[3]):
                raise ValueError(f"legacy dims {(b, s, h, d)} do not match q.shape {tuple(q.shape)}")
        else:
            raise TypeError("flash_attn_func(q, k, v[, batch_size, seq_len, num_heads, head_dim], causal=False)")
    if return_lse:
        out, lse = _C.fwd(q, k, v, bool(causal))
        return out, lse
    return _C.attn_autograd(q, k, v, bool(causal))


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal=False):
    return _C.attn_varlen_autograd(q, k, v, cu_seqlens_q, cu_seqlens_k, int(max_seqlen_q), int(max_seqlen_k), bool(causal))
