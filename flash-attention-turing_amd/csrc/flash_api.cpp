// flash_api.cpp — PyTorch host module `flash_attn_turing._C` for the MI355X attention library.
//
// Same Python surface as the reference extension (csrc/flash_attn/flash_api.cpp:471-476):
//     fwd(q, k, v, is_causal)                               -> [o, l]
//     bwd(q, k, v, out, l, dout, is_causal)                 -> [dq, dk, dv]
//     varlen_fwd(q, k, v, cu_q, cu_k, max_sq, max_sk, is_causal)            -> [out, l]
//     varlen_bwd(q, k, v, out, l, dout, cu_q, cu_k, max_sq, max_sk, causal) -> [dq, dk, dv]
// Shape checks and their messages follow the reference's TORCH_CHECKs (:178-183, :252-259,
// :329-345, :396-423).  Deliberate hardening over the reference: dtype / device /
// head_dim checks (the reference silently returns zeros for head_dim not in {64,128},
// static_switch.h:29-38), bf16 accepted in addition to fp16, kernels enqueued on the CURRENT
// stream of q's device under a device guard (the reference uses the legacy default stream of
// device 0), launch errors surfaced.  This file contains no device code: it is compiled by
// plain g++ and calls the C ABI in include/flash_attn_gfx950.h; PyTorch is only plumbing
// (allocation, streams).
#include <atomic>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include "flash_attn_gfx950.h"

namespace {

int fa_dtype_of(const at::Tensor& t) {
    if (t.scalar_type() == at::kHalf) return FA_FP16;
    if (t.scalar_type() == at::kBFloat16) return FA_BF16;
    TORCH_CHECK(false, "flash_attn_turing: only fp16 and bf16 are supported, got ", t.scalar_type());
    return -1;
}

void check_qkv_common(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v) {
    TORCH_CHECK(q.is_cuda() && k.is_cuda() && v.is_cuda(), "q, k, v must be GPU (HIP) tensors");
    TORCH_CHECK(k.device() == q.device() && v.device() == q.device(), "q, k, v must be on the same device");
    TORCH_CHECK(k.scalar_type() == q.scalar_type() && v.scalar_type() == q.scalar_type(), "q, k, v must have the same dtype");
}

void check_status(int rc) {
    TORCH_CHECK(rc == FA_OK, "flash_attn_turing (gfx950): ", fa_last_error(), " [code ", rc, "]");
}

void* current_stream(const at::Tensor& t) {
    return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

// (b, s, h, d) tensor -> element strides; the innermost dim must be dense.
fa_strides strides4(const at::Tensor& t) {
    TORCH_CHECK(t.stride(3) == 1, "last dimension must be contiguous");
    return fa_strides{t.stride(0), t.stride(1), t.stride(2)};
}
fa_strides strides3(const at::Tensor& t) {
    TORCH_CHECK(t.stride(2) == 1, "last dimension must be contiguous");
    return fa_strides{0, t.stride(0), t.stride(1)};
}
// The kernels take real strides, but rows/heads must stay 16-byte aligned; anything else is
// densified (the reference assumes contiguous input without checking, flash_api.cpp:38-54).
// NOTE: the densifying copy is a hidden extra HBM pass the reference never makes (it never checks); it only
// triggers for layouts the kernels cannot address (unaligned base / strides, broadcast rows) and is counted in
// g_densify_copies (exported as `densify_copies()`) so that a caller can see it happened.
std::atomic<int64_t> g_densify_copies{0};       // autograd may run backward on several threads / devices at once
at::Tensor dense_last(const at::Tensor& t) {
    bool ok = t.stride(-1) == 1 && (reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0);
    for (int i = 0; i < t.dim() - 1 && ok; ++i) ok = (t.stride(i) % 8 == 0);
    // the row (sequence) dimension is dim -3: an expand()-ed / overlapping row stride (< head_dim) is not addressable
    if (ok && t.dim() >= 3 && t.size(-3) > 1) ok = t.stride(-3) >= t.size(-1);
    if (ok) return t;
    g_densify_copies.fetch_add(1, std::memory_order_relaxed);
    return t.contiguous();
}
void check_same_device(const at::Tensor& q, const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.device() == q.device(), name, " must be on the same GPU device as q");
}

void check_cu_seqlens(const at::Tensor& cu_q, const at::Tensor& cu_k) {
    TORCH_CHECK(cu_q.is_cuda() && cu_k.is_cuda(), "cu_seqlens_q/cu_seqlens_k must be CUDA tensors");
    TORCH_CHECK(cu_q.scalar_type() == torch::kInt32 && cu_k.scalar_type() == torch::kInt32,
                "cu_seqlens_q/cu_seqlens_k must be int32 tensors");
    TORCH_CHECK(cu_q.is_contiguous() && cu_k.is_contiguous(), "cu_seqlens_q/cu_seqlens_k must be contiguous");
    TORCH_CHECK(cu_q.dim() == 1 && cu_k.dim() == 1, "cu_seqlens_q/cu_seqlens_k must be rank-1");
    TORCH_CHECK(cu_q.numel() >= 2 && cu_k.numel() >= 2, "cu_seqlens_q/cu_seqlens_k must have at least 2 elements");
    TORCH_CHECK(cu_k.numel() == cu_q.numel(), "cu_seqlens_k must have shape [batch_size + 1] with cumulative offsets");
}

}  // namespace

// reference: mha_fwd, flash_api.cpp:156-223
std::vector<at::Tensor> mha_fwd(at::Tensor q, at::Tensor k, at::Tensor v, bool is_causal) {
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must be rank-4 tensors");
    check_qkv_common(q, k, v);
    const int64_t batch_size = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_size = q.size(3);
    const int64_t seqlen_k = k.size(1), num_heads_k = k.size(2);
    TORCH_CHECK(k.size(0) == batch_size && v.size(0) == batch_size, "k/v batch size must match q");
    TORCH_CHECK(v.size(1) == seqlen_k, "k and v seqlen_k must match");
    TORCH_CHECK(v.size(2) == num_heads_k, "k and v num_heads must match");
    TORCH_CHECK(k.size(3) == head_size && v.size(3) == head_size, "q/k/v head_dim must match");
    TORCH_CHECK(num_heads_k > 0 && num_heads % num_heads_k == 0, "num_heads_q must be divisible by num_heads_k for GQA/MQA");

    c10::DeviceGuard guard(q.device());   // resolves to the ROCm (cuda-masquerading) guard impl
    q = dense_last(q); k = dense_last(k); v = dense_last(v);
    // every element of o and l is written by the kernel (dead rows included), so no zero fill
    at::Tensor o = torch::empty(q.sizes(), q.options());
    at::Tensor l = torch::empty({batch_size, num_heads, seqlen_q}, q.options().dtype(torch::kFloat32));

    fa_fwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr(); p.o = o.data_ptr(); p.lse = l.data_ptr<float>();
    p.b = (int32_t)batch_size; p.seqlen_q = (int32_t)seqlen_q; p.seqlen_k = (int32_t)seqlen_k;
    p.h = (int32_t)num_heads; p.h_k = (int32_t)num_heads_k; p.d = (int32_t)head_size;
    p.dtype = fa_dtype_of(q); p.is_causal = is_causal;
    p.q_stride = strides4(q); p.k_stride = strides4(k); p.v_stride = strides4(v); p.o_stride = strides4(o);
    check_status(fa_run_mha_fwd(&p, current_stream(q)));
    return {o, l};
}

// reference: mha_bwd, flash_api.cpp:228-317
std::vector<at::Tensor> mha_bwd(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor out, at::Tensor l, at::Tensor dout,
                                bool is_causal) {
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must be rank-4 tensors");
    TORCH_CHECK(out.dim() == 4 && dout.dim() == 4, "out and dout must be rank-4 tensors");
    check_qkv_common(q, k, v);
    check_same_device(q, out, "out"); check_same_device(q, dout, "dout"); check_same_device(q, l, "l");
    const int64_t batch_size = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_size = q.size(3);
    const int64_t seqlen_k = k.size(1), num_heads_k = k.size(2);
    TORCH_CHECK(k.size(0) == batch_size && v.size(0) == batch_size, "k/v batch size must match q");
    TORCH_CHECK(v.size(1) == seqlen_k, "k and v seqlen_k must match");
    TORCH_CHECK(v.size(2) == num_heads_k, "k and v num_heads must match");
    TORCH_CHECK(k.size(3) == head_size && v.size(3) == head_size, "q/k/v head_dim must match");
    TORCH_CHECK(out.sizes() == q.sizes() && dout.sizes() == q.sizes(), "out and dout must match q shape");
    TORCH_CHECK(num_heads_k > 0 && num_heads % num_heads_k == 0, "num_heads_q must be divisible by num_heads_k for GQA/MQA");
    TORCH_CHECK(out.scalar_type() == q.scalar_type() && dout.scalar_type() == q.scalar_type(), "out/dout dtype must match q");
    TORCH_CHECK(l.scalar_type() == torch::kFloat32 && l.dim() == 3 && l.size(0) == batch_size && l.size(1) == num_heads &&
                    l.size(2) == seqlen_q, "l must be fp32 with shape [batch_size, nheads_q, seqlen_q]");

    c10::DeviceGuard guard(q.device());   // resolves to the ROCm (cuda-masquerading) guard impl
    q = dense_last(q); k = dense_last(k); v = dense_last(v); out = dense_last(out); dout = dense_last(dout);
    l = l.contiguous();
    at::Tensor dq = torch::empty(q.sizes(), q.options());
    at::Tensor dk = torch::empty(k.sizes(), k.options());
    at::Tensor dv = torch::empty(v.sizes(), v.options());
    at::Tensor do_o = torch::empty_like(l);   // D = rowsum(dO * O), the reference's do_o (:274)

    fa_bwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr(); p.o = out.data_ptr(); p.dout = dout.data_ptr();
    p.lse = l.data_ptr<float>(); p.dsoftmax_sum = do_o.data_ptr<float>();
    p.dq = dq.data_ptr(); p.dk = dk.data_ptr(); p.dv = dv.data_ptr();
    p.b = (int32_t)batch_size; p.seqlen_q = (int32_t)seqlen_q; p.seqlen_k = (int32_t)seqlen_k;
    p.h = (int32_t)num_heads; p.h_k = (int32_t)num_heads_k; p.d = (int32_t)head_size;
    p.dtype = fa_dtype_of(q); p.is_causal = is_causal;
    p.q_stride = strides4(q); p.k_stride = strides4(k); p.v_stride = strides4(v); p.o_stride = strides4(out);
    p.do_stride = strides4(dout); p.dq_stride = strides4(dq); p.dk_stride = strides4(dk); p.dv_stride = strides4(dv);
    // fp32 scratch (ABI 3) that lets the dK/dV launch split a KV head's query-head group over several workgroups (GQA / MQA with few
    // workgroups, causal imbalance); 0 bytes for MHA and for grids that fill the chip anyway
    at::Tensor workspace;
    const int64_t ws_bytes = fa_bwd_workspace_bytes(&p);
    if (ws_bytes < 0) check_status((int)ws_bytes);
    if (ws_bytes > 0) {
        workspace = torch::empty({ws_bytes / 4}, q.options().dtype(torch::kFloat32));
        p.workspace = workspace.data_ptr(); p.workspace_bytes = ws_bytes;
    }
    check_status(fa_run_mha_bwd(&p, current_stream(q)));
    return {dq, dk, dv};
}

// reference: mha_varlen_fwd, flash_api.cpp:319-381
std::vector<at::Tensor> mha_varlen_fwd(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor& cu_seqlens_q,
                                       at::Tensor& cu_seqlens_k, const int max_seqlen_q, const int max_seqlen_k,
                                       bool is_causal) {
    TORCH_CHECK(q.dim() == 3 && k.dim() == 3 && v.dim() == 3, "q, k, v must be rank-3 packed tensors");
    check_cu_seqlens(cu_seqlens_q, cu_seqlens_k);
    check_qkv_common(q, k, v);
    check_same_device(q, cu_seqlens_q, "cu_seqlens_q"); check_same_device(q, cu_seqlens_k, "cu_seqlens_k");
    const int64_t batch_size = cu_seqlens_q.numel() - 1;
    TORCH_CHECK(k.size(0) == v.size(0), "k and v total tokens must match");
    TORCH_CHECK(k.size(1) == v.size(1), "k and v num_heads must match");
    TORCH_CHECK(k.size(2) == v.size(2), "k and v head_dim must match");
    TORCH_CHECK(q.size(2) == k.size(2), "q/k/v head_dim must match");
    TORCH_CHECK(k.size(1) > 0 && q.size(1) % k.size(1) == 0, "num_heads_q must be divisible by num_heads_k for GQA/MQA");
    TORCH_CHECK(max_seqlen_q >= 0 && max_seqlen_k >= 0, "max_seqlen_q/max_seqlen_k must be non-negative");
    const int64_t num_heads = q.size(1), num_heads_k = k.size(1), head_size = q.size(2);

    c10::DeviceGuard guard(q.device());   // resolves to the ROCm (cuda-masquerading) guard impl
    q = dense_last(q); k = dense_last(k); v = dense_last(v);
    // tokens past cu_seqlens_q[-1] belong to no sequence and are not touched by the kernel: keep
    // the reference's zero fill (flash_api.cpp:351) for them
    at::Tensor out = torch::zeros_like(q);
    // padded LSE: entries past a sequence's length are never written by the kernel -> keep the
    // reference's zero fill for those (flash_api.cpp:352)
    at::Tensor l = torch::zeros({batch_size, num_heads, max_seqlen_q}, q.options().dtype(torch::kFloat32));

    fa_fwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr(); p.o = out.data_ptr(); p.lse = l.data_ptr<float>();
    p.cu_seqlens_q = cu_seqlens_q.data_ptr<int32_t>(); p.cu_seqlens_k = cu_seqlens_k.data_ptr<int32_t>();
    p.b = (int32_t)batch_size; p.seqlen_q = max_seqlen_q; p.seqlen_k = max_seqlen_k;
    p.h = (int32_t)num_heads; p.h_k = (int32_t)num_heads_k; p.d = (int32_t)head_size;
    p.dtype = fa_dtype_of(q); p.is_causal = is_causal;
    p.q_stride = strides3(q); p.k_stride = strides3(k); p.v_stride = strides3(v); p.o_stride = strides3(out);
    p.total_q = q.size(0); p.total_k = k.size(0);   // packed row counts >= cu_seqlens[b]: lets the library size the grid by tokens present
    check_status(fa_run_mha_fwd(&p, current_stream(q)));
    return {out, l};
}

// reference: mha_varlen_bwd, flash_api.cpp:383-468
std::vector<at::Tensor> mha_varlen_bwd(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor out, at::Tensor l,
                                       at::Tensor dout, at::Tensor cu_seqlens_q, at::Tensor cu_seqlens_k,
                                       const int max_seqlen_q, const int max_seqlen_k, bool is_causal) {
    TORCH_CHECK(q.dim() == 3 && k.dim() == 3 && v.dim() == 3, "q, k, v must be rank-3 packed tensors");
    check_cu_seqlens(cu_seqlens_q, cu_seqlens_k);
    check_qkv_common(q, k, v);
    check_same_device(q, cu_seqlens_q, "cu_seqlens_q"); check_same_device(q, cu_seqlens_k, "cu_seqlens_k");
    check_same_device(q, out, "out"); check_same_device(q, dout, "dout"); check_same_device(q, l, "l");
    const int64_t batch_size = cu_seqlens_q.numel() - 1;
    TORCH_CHECK(k.size(0) == v.size(0), "k and v total tokens must match");
    TORCH_CHECK(k.size(1) == v.size(1), "k and v num_heads must match");
    TORCH_CHECK(k.size(2) == v.size(2), "k and v head_dim must match");
    TORCH_CHECK(q.size(2) == k.size(2), "q/k/v head_dim must match");
    TORCH_CHECK(k.size(1) > 0 && q.size(1) % k.size(1) == 0, "num_heads_q must be divisible by num_heads_k for GQA/MQA");
    TORCH_CHECK(out.sizes() == q.sizes(), "out must match q shape");
    TORCH_CHECK(dout.sizes() == q.sizes(), "dout must match q shape");
    TORCH_CHECK(l.dim() == 3, "l must be rank-3 for varlen_bwd");
    const int64_t num_heads = q.size(1), num_heads_k = k.size(1), head_size = q.size(2);
    TORCH_CHECK(l.size(0) == batch_size && l.size(1) == num_heads && l.size(2) == max_seqlen_q,
                "l must have shape [batch_size, nheads_q, max_seqlen_q]");
    TORCH_CHECK(l.scalar_type() == torch::kFloat32, "l must be fp32");
    TORCH_CHECK(out.scalar_type() == q.scalar_type() && dout.scalar_type() == q.scalar_type(), "out/dout dtype must match q");

    c10::DeviceGuard guard(q.device());   // resolves to the ROCm (cuda-masquerading) guard impl
    q = dense_last(q); k = dense_last(k); v = dense_last(v); out = dense_last(out); dout = dense_last(dout);
    l = l.contiguous();
    at::Tensor dq = torch::zeros_like(q);
    // tokens of k/v that belong to no sequence (beyond cu_seqlens_k[-1]) are not touched by
    // the kernels -> zero fill like the reference (:425-427)
    at::Tensor dk = torch::zeros_like(k);
    at::Tensor dv = torch::zeros_like(v);
    at::Tensor do_o = torch::zeros_like(l);

    fa_bwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr(); p.o = out.data_ptr(); p.dout = dout.data_ptr();
    p.lse = l.data_ptr<float>(); p.dsoftmax_sum = do_o.data_ptr<float>();
    p.dq = dq.data_ptr(); p.dk = dk.data_ptr(); p.dv = dv.data_ptr();
    p.cu_seqlens_q = cu_seqlens_q.data_ptr<int32_t>(); p.cu_seqlens_k = cu_seqlens_k.data_ptr<int32_t>();
    p.b = (int32_t)batch_size; p.seqlen_q = max_seqlen_q; p.seqlen_k = max_seqlen_k;
    p.h = (int32_t)num_heads; p.h_k = (int32_t)num_heads_k; p.d = (int32_t)head_size;
    p.dtype = fa_dtype_of(q); p.is_causal = is_causal;
    p.q_stride = strides3(q); p.k_stride = strides3(k); p.v_stride = strides3(v); p.o_stride = strides3(out);
    p.do_stride = strides3(dout); p.dq_stride = strides3(dq); p.dk_stride = strides3(dk); p.dv_stride = strides3(dv);
    p.total_q = q.size(0); p.total_k = k.size(0);
    // fp32 scratch (ABI 3) that lets the dK/dV launch split a KV head's query-head group over several workgroups (GQA / MQA with few
    // workgroups, causal imbalance); 0 bytes for MHA and for grids that fill the chip anyway
    at::Tensor workspace;
    const int64_t ws_bytes = fa_bwd_workspace_bytes(&p);
    if (ws_bytes < 0) check_status((int)ws_bytes);
    if (ws_bytes > 0) {
        workspace = torch::empty({ws_bytes / 4}, q.options().dtype(torch::kFloat32));
        p.workspace = workspace.data_ptr(); p.workspace_bytes = ws_bytes;
    }
    check_status(fa_run_mha_bwd(&p, current_stream(q)));
    return {dq, dk, dv};
}

// ---- autograd nodes in C++ ------------------------------------------------------------------------------------------------------
// The reference ships the four raw functions only; its README's flash_attn_func is an older Python-level API.  The differentiable wrappers
// of this package used to be torch.autograd.Function subclasses in Python: ~85 us of host time per forward + backward, more than the GPU
// work of a b1 x 512-token call (58 us).  As C++ nodes the same pair costs the host 42-52 us (tools/host_overhead.py, profiles/r4_host_overhead.log).
class FlashAttnNode : public torch::autograd::Function<FlashAttnNode> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor q, at::Tensor k, at::Tensor v, bool is_causal) {
        at::AutoDispatchBelowADInplaceOrView below;
        std::vector<at::Tensor> r = mha_fwd(q, k, v, is_causal);
        ctx->save_for_backward({q, k, v, r[0], r[1]});
        ctx->saved_data["causal"] = is_causal;
        return r[0];
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list grads) {
        const torch::autograd::variable_list s = ctx->get_saved_variables();
        std::vector<at::Tensor> g = mha_bwd(s[0], s[1], s[2], s[3], s[4], grads[0], ctx->saved_data["causal"].toBool());      // strided dout is fine
        return {g[0], g[1], g[2], at::Tensor()};
    }
};
class FlashAttnVarlenNode : public torch::autograd::Function<FlashAttnVarlenNode> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor cu_seqlens_q, at::Tensor cu_seqlens_k,
                              int64_t max_seqlen_q, int64_t max_seqlen_k, bool is_causal) {
        at::AutoDispatchBelowADInplaceOrView below;
        std::vector<at::Tensor> r = mha_varlen_fwd(q, k, v, cu_seqlens_q, cu_seqlens_k, (int)max_seqlen_q, (int)max_seqlen_k, is_causal);
        ctx->save_for_backward({q, k, v, r[0], r[1], cu_seqlens_q, cu_seqlens_k});
        ctx->saved_data["causal"] = is_causal;
        ctx->saved_data["max_q"] = max_seqlen_q;
        ctx->saved_data["max_k"] = max_seqlen_k;
        return r[0];
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list grads) {
        const torch::autograd::variable_list s = ctx->get_saved_variables();
        std::vector<at::Tensor> g = mha_varlen_bwd(s[0], s[1], s[2], s[3], s[4], grads[0], s[5], s[6], (int)ctx->saved_data["max_q"].toInt(),
                                                   (int)ctx->saved_data["max_k"].toInt(), ctx->saved_data["causal"].toBool());
        return {g[0], g[1], g[2], at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};
static at::Tensor attn_autograd(at::Tensor q, at::Tensor k, at::Tensor v, bool is_causal) { return FlashAttnNode::apply(q, k, v, is_causal); }
static at::Tensor attn_varlen_autograd(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor cu_seqlens_q, at::Tensor cu_seqlens_k, int64_t max_seqlen_q,
                                       int64_t max_seqlen_k, bool is_causal) {
    return FlashAttnVarlenNode::apply(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, is_causal);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "MI355X (gfx950) fused attention behind the flash_attn_turing surface";
    m.def("fwd", &mha_fwd, "Forward pass");
    m.def("bwd", &mha_bwd, "Backward pass");
    m.def("varlen_fwd", &mha_varlen_fwd, "Varlen forward pass");
    m.def("varlen_bwd", &mha_varlen_bwd, "Varlen backward pass");
    m.def("attn_autograd", &attn_autograd, "differentiable forward (C++ autograd node over fwd / bwd)");
    m.def("attn_varlen_autograd", &attn_varlen_autograd, "differentiable packed forward (C++ autograd node over varlen_fwd / varlen_bwd)");
    m.def("abi_version", []() { return fa_abi_version(); });
    m.def("build_info", []() { return std::string(fa_build_info()); });
    m.def("densify_copies", []() { return g_densify_copies.load(std::memory_order_relaxed); }, "number of hidden .contiguous() copies made so far (0 for addressable layouts)");
}
