// fa_capi.hip — the C-ABI boundary (include/flash_attn_gfx950.h).
// Host-side validation + parameter fill + dispatch; the counterpart of the reference's
// set_params_fprop / set_params_dgrad / run_mha_fwd / run_mha_bwd
// (csrc/flash_attn/flash_api.cpp:5-153) without any torch type in sight.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "fa_params.hpp"
#include "flash_attn_gfx950.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

fa::TStride conv(const fa_strides& s) { return fa::TStride{s.batch, s.row, s.head}; }

fa_strides contiguous_bshd(int64_t s, int64_t h, int64_t d) { return fa_strides{s * h * d, h * d, d}; }

// Rows must be 16-byte aligned for the 128-bit loads; one sequence of one batch entry must fit
// a 32-bit byte offset (the kernels address rows through buffer descriptors).
int check_tensor(const char* name, const void* ptr, const fa_strides& st, int64_t rows, int d, bool varlen) {
    if (ptr == nullptr) return fail(FA_ERR_NULL_POINTER, "%s is NULL", name);
    if (((uintptr_t)ptr & 15) != 0) return fail(FA_ERR_BAD_STRIDE, "%s base pointer must be 16-byte aligned", name);
    if (st.row < d || st.head < 0 || (st.row % 8) != 0 || (st.head % 8) != 0 || (!varlen && (st.batch % 8) != 0))
        return fail(FA_ERR_BAD_STRIDE, "%s strides (batch=%lld,row=%lld,head=%lld) must be multiples of 8 elements with row >= head_dim",
                    name, (long long)st.batch, (long long)st.row, (long long)st.head);
    if (rows * st.row * 2 >= ((int64_t)1 << 31))
        return fail(FA_ERR_BAD_STRIDE, "%s: one sequence spans %lld bytes (limit 2^31)", name, (long long)(rows * st.row * 2));
    return FA_OK;
}

int check_common(int b, int sq, int sk, int h, int hk, int d, int dtype) {
    if (b < 0 || sq < 0 || sk < 0 || h <= 0 || hk <= 0)
        return fail(FA_ERR_BAD_SHAPE, "bad sizes b=%d seqlen_q=%d seqlen_k=%d h=%d h_k=%d", b, sq, sk, h, hk);
    if (h % hk != 0) return fail(FA_ERR_BAD_GQA, "num_heads_q (%d) must be divisible by num_heads_k (%d) for GQA/MQA", h, hk);
    if (d != 64 && d != 128) return fail(FA_ERR_BAD_HEADDIM, "head_dim %d unsupported (64 or 128)", d);
    if (dtype != FA_FP16 && dtype != FA_BF16) return fail(FA_ERR_BAD_DTYPE, "dtype %d unsupported (0=fp16, 1=bf16)", dtype);
    return FA_OK;
}

// ABI 4: the caller's struct -> a zeroed local one, min(caller's size, ours) bytes.  Fields appended after the caller's header was
// written stay 0 / NULL ("not given"); a struct without the {size, magic} header (ABI 1-3 callers) or longer than ours is an error.
template <typename P>
int import_params(const P* user, P& local, const char* what) {
    if (user == nullptr) return fail(FA_ERR_NULL_POINTER, "params is NULL");
    if (user->magic != FA_PARAMS_MAGIC)
        return fail(FA_ERR_BAD_ABI, "%s: no {struct_size, magic} header - caller was compiled against an ABI < 4 header; recompile against include/flash_attn_gfx950.h (ABI %d)",
                    what, FA_ABI_VERSION);
    const size_t base = offsetof(P, total_q);           // everything up to the first appended (optional) field is mandatory
    if (user->struct_size < base || user->struct_size > sizeof(P))
        return fail(FA_ERR_BAD_ABI, "%s: struct_size %u outside [%zu, %zu] - header / library mismatch", what, user->struct_size, base, sizeof(P));
    memset(&local, 0, sizeof(P));
    memcpy(&local, user, user->struct_size);
    return FA_OK;
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return FA_OK;
    fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

}  // namespace

extern "C" {

int fa_abi_version(void) { return FA_ABI_VERSION; }
const char* fa_last_error(void) { return g_err; }

#define FA_STR2(x) #x
#define FA_STR(x) FA_STR2(x)
#ifndef FA_SOURCE_DIGEST
#define FA_SOURCE_DIGEST "unstamped"      // build.py passes the first 12 hex digits of the sha256 over kernel sources, headers and flags
#endif
const char* fa_build_info(void) { return "flash_attn_gfx950 abi=" FA_STR(FA_ABI_VERSION) " arch=gfx950 mfma=32x32x16+16x16x32 wave64 src=" FA_SOURCE_DIGEST " built " __DATE__; }

double fa_fwd_flops(int32_t b, int32_t sq, int32_t sk, int32_t h, int32_t d, int32_t is_causal) {
    double pairs;
    if (!is_causal) {
        pairs = (double)sq * (double)sk;
    } else {
        // visible (i, j) pairs with j <= i + (sk - sq), 0 <= j < sk
        const int64_t delta = (int64_t)sk - sq;
        double acc = 0.0;
        // rows i with i + delta < 0 see nothing; rows with i + delta >= sk-1 see sk
        int64_t i0 = delta < 0 ? -delta : 0;        // first row that sees key 0
        if (i0 < sq) {
            int64_t n = sq - i0;                     // rows i0..sq-1 see (i + delta + 1) keys (<= sk always since i<=sq-1)
            double first = (double)(i0 + delta + 1), last = (double)(sq - 1 + delta + 1);
            acc = (first + last) * (double)n / 2.0;
        }
        pairs = acc;
    }
    return 4.0 * (double)b * (double)h * pairs * (double)d;
}

double fa_fwd_bytes(int32_t b, int32_t sq, int32_t sk, int32_t h, int32_t hk, int32_t d) {
    return 2.0 * ((double)b * sq * h * d * 2.0 + (double)b * sk * hk * d * 2.0) + (double)b * h * sq * 4.0;
}

const char* fa_fwd_kernel_name(int32_t d) { return fa::fwd_kernel_name(d); }
int32_t fa_set_kernel_policy(int32_t policy) { return fa::set_kernel_policy(policy); }
int64_t fa_set_policy_problem_heads(int64_t batch_times_heads) { return fa::set_policy_problem_heads(batch_times_heads); }
const char* fa_kernel_name(int32_t stage, int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t d, int32_t is_causal) {
    return fa_kernel_name_dtype(stage, FA_FP16, b, seqlen_q, seqlen_k, h, d, is_causal);
}
const char* fa_kernel_name_dtype(int32_t stage, int32_t dtype, int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t d, int32_t is_causal) {
    if (dtype != FA_FP16 && dtype != FA_BF16) return "";
    if (stage == FA_STAGE_FWD) {
        fa::FwdKernelParams kp{};
        kp.b = b; kp.seqlen_q = seqlen_q; kp.seqlen_k = seqlen_k; kp.h = h; kp.d = d; kp.is_causal = is_causal;
        return fa::fwd_kernel_name_for(kp, dtype);
    }
    if (stage != FA_STAGE_DQ && stage != FA_STAGE_DKDV) return "";
    fa::BwdKernelParams kp{};
    kp.b = b; kp.seqlen_q = seqlen_q; kp.seqlen_k = seqlen_k; kp.h = h; kp.d = d; kp.is_causal = is_causal;
    return fa::bwd_kernel_name_for(kp, stage == FA_STAGE_DKDV);
}

int fa_device_clock_khz(int32_t device) {
    int khz = 0;
    const hipError_t e = hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, device);
    return e == hipSuccess ? khz : -(int)e;
}

int fa_run_mha_fwd(const fa_fwd_params* user, void* stream) {
    fa_fwd_params local;
    int rc = import_params(user, local, "fa_fwd_params");
    if (rc) return rc;
    const fa_fwd_params* p = &local;
    rc = check_common(p->b, p->seqlen_q, p->seqlen_k, p->h, p->h_k, p->d, p->dtype);
    if (rc) return rc;
    const bool varlen = p->cu_seqlens_q != nullptr || p->cu_seqlens_k != nullptr;
    if (varlen && (p->cu_seqlens_q == nullptr || p->cu_seqlens_k == nullptr))
        return fail(FA_ERR_NULL_POINTER, "cu_seqlens_q and cu_seqlens_k must both be given for varlen");
    if (p->b == 0 || p->seqlen_q == 0) return FA_OK;   // nothing to write
    if (p->lse == nullptr) return fail(FA_ERR_NULL_POINTER, "lse is NULL");
    if ((rc = check_tensor("q", p->q, p->q_stride, p->seqlen_q, p->d, varlen))) return rc;
    if ((rc = check_tensor("o", p->o, p->o_stride, p->seqlen_q, p->d, varlen))) return rc;
    if (p->seqlen_k > 0) {
        if ((rc = check_tensor("k", p->k, p->k_stride, p->seqlen_k, p->d, varlen))) return rc;
        if ((rc = check_tensor("v", p->v, p->v_stride, p->seqlen_k, p->d, varlen))) return rc;
    }
    fa::FwdKernelParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.q_ptr = p->q; kp.k_ptr = p->k; kp.v_ptr = p->v; kp.o_ptr = p->o; kp.lse_ptr = p->lse;
    kp.cu_seqlens_q = p->cu_seqlens_q; kp.cu_seqlens_k = p->cu_seqlens_k;
    kp.q = conv(p->q_stride); kp.k = conv(p->k_stride); kp.v = conv(p->v_stride); kp.o = conv(p->o_stride);
    kp.lse_row_stride = p->seqlen_q;
    kp.b = p->b; kp.seqlen_q = p->seqlen_q; kp.seqlen_k = p->seqlen_k;
    kp.h = p->h; kp.h_k = p->h_k; kp.h_ratio = p->h / p->h_k; kp.d = p->d;
    kp.is_causal = p->is_causal ? 1 : 0;
    if (p->total_q < 0 || p->total_k < 0) return fail(FA_ERR_BAD_SHAPE, "total_q / total_k must be >= 0 (0 = unknown)");
    kp.total_q = varlen ? p->total_q : 0;          // sizes the varlen launch grid by the tokens present (fa_device.hpp)
    kp.scale = 1.0f / sqrtf((float)p->d);          // hard-wired like the reference (flash_fwd_kernel.h:351)
    kp.scale_log2e = kp.scale * 1.4426950408889634f;
    return hip_status(fa::launch_fwd(kp, p->dtype, (hipStream_t)stream), "fa_fwd launch");
}

// with_workspace = false: the `workspace` fields are not looked at (fa_bwd_workspace_bytes, and the launches that never use it)
static int fill_bwd(const fa_bwd_params* user, fa::BwdKernelParams& kp, fa_bwd_params& local, bool with_workspace) {
    int rc = import_params(user, local, "fa_bwd_params");
    if (rc) return rc;
    const fa_bwd_params* p = &local;
    rc = check_common(p->b, p->seqlen_q, p->seqlen_k, p->h, p->h_k, p->d, p->dtype);
    if (rc) return rc;
    const bool varlen = p->cu_seqlens_q != nullptr || p->cu_seqlens_k != nullptr;
    if (varlen && (p->cu_seqlens_q == nullptr || p->cu_seqlens_k == nullptr))
        return fail(FA_ERR_NULL_POINTER, "cu_seqlens_q and cu_seqlens_k must both be given for varlen");
    if (p->b > 0 && p->seqlen_q > 0) {
        if (p->lse == nullptr || p->dsoftmax_sum == nullptr) return fail(FA_ERR_NULL_POINTER, "lse / dsoftmax_sum is NULL");
        if ((rc = check_tensor("q", p->q, p->q_stride, p->seqlen_q, p->d, varlen))) return rc;
        if ((rc = check_tensor("o", p->o, p->o_stride, p->seqlen_q, p->d, varlen))) return rc;
        if ((rc = check_tensor("dout", p->dout, p->do_stride, p->seqlen_q, p->d, varlen))) return rc;
        if ((rc = check_tensor("dq", p->dq, p->dq_stride, p->seqlen_q, p->d, varlen))) return rc;
    }
    if (p->b > 0 && p->seqlen_k > 0) {
        if ((rc = check_tensor("k", p->k, p->k_stride, p->seqlen_k, p->d, varlen))) return rc;
        if ((rc = check_tensor("v", p->v, p->v_stride, p->seqlen_k, p->d, varlen))) return rc;
        if ((rc = check_tensor("dk", p->dk, p->dk_stride, p->seqlen_k, p->d, varlen))) return rc;
        if ((rc = check_tensor("dv", p->dv, p->dv_stride, p->seqlen_k, p->d, varlen))) return rc;
    }
    memset(&kp, 0, sizeof(kp));
    kp.q_ptr = p->q; kp.k_ptr = p->k; kp.v_ptr = p->v; kp.o_ptr = p->o; kp.do_ptr = p->dout;
    kp.lse_ptr = p->lse; kp.dsum_ptr = p->dsoftmax_sum;
    kp.dq_ptr = p->dq; kp.dk_ptr = p->dk; kp.dv_ptr = p->dv;
    kp.cu_seqlens_q = p->cu_seqlens_q; kp.cu_seqlens_k = p->cu_seqlens_k;
    kp.q = conv(p->q_stride); kp.k = conv(p->k_stride); kp.v = conv(p->v_stride); kp.o = conv(p->o_stride);
    kp.dout = conv(p->do_stride); kp.dq = conv(p->dq_stride); kp.dk = conv(p->dk_stride); kp.dv = conv(p->dv_stride);
    kp.lse_row_stride = p->seqlen_q;
    kp.b = p->b; kp.seqlen_q = p->seqlen_q; kp.seqlen_k = p->seqlen_k;
    kp.h = p->h; kp.h_k = p->h_k; kp.h_ratio = p->h / p->h_k; kp.d = p->d;
    kp.is_causal = p->is_causal ? 1 : 0;
    if (p->total_q < 0 || p->total_k < 0) return fail(FA_ERR_BAD_SHAPE, "total_q / total_k must be >= 0 (0 = unknown)");
    if (p->cu_seqlens_q != nullptr) { kp.total_q = p->total_q; kp.total_k = p->total_k; }
    kp.scale = 1.0f / sqrtf((float)p->d);
    kp.scale_log2e = kp.scale * 1.4426950408889634f;
    if (with_workspace && p->workspace_bytes < 0) return fail(FA_ERR_BAD_SHAPE, "workspace_bytes must be >= 0");
    if (with_workspace && p->workspace != nullptr && p->workspace_bytes > 0) {
        if ((reinterpret_cast<uintptr_t>(p->workspace) & 15u) != 0) return fail(FA_ERR_BAD_STRIDE, "workspace must be 16-byte aligned");
        kp.ws = (float*)p->workspace; kp.ws_bytes = p->workspace_bytes;
    }
    kp.n_split = 1;
    return FA_OK;
}

int64_t fa_bwd_workspace_bytes(const fa_bwd_params* user) {
    fa::BwdKernelParams kp;
    fa_bwd_params local;
    int rc = fill_bwd(user, kp, local, false);
    if (rc) return rc;
    const fa_bwd_params* p = &local;
    if (p->b == 0 || p->seqlen_k == 0 || p->seqlen_q == 0) return 0;
    return fa::dkdv_workspace_bytes(kp, fa::dkdv_split(kp, -1));
}

int fa_bwd_dot_do_o(const fa_bwd_params* user, void* stream) {
    fa::BwdKernelParams kp;
    fa_bwd_params local;
    int rc = fill_bwd(user, kp, local, false);
    if (rc) return rc;
    const fa_bwd_params* p = &local;
    if (p->b == 0 || p->seqlen_q == 0) return FA_OK;
    return hip_status(fa::launch_bwd_dot_do_o(kp, p->dtype, (hipStream_t)stream), "fa_bwd_dot_do_o launch");
}

int fa_bwd_dq(const fa_bwd_params* user, void* stream) {
    fa::BwdKernelParams kp;
    fa_bwd_params local;
    int rc = fill_bwd(user, kp, local, false);
    if (rc) return rc;
    const fa_bwd_params* p = &local;
    if (p->b == 0 || p->seqlen_q == 0) return FA_OK;
    return hip_status(fa::launch_bwd_dq(kp, p->dtype, (hipStream_t)stream), "fa_bwd_dq launch");
}

int fa_bwd_dkdv(const fa_bwd_params* user, void* stream) {
    fa::BwdKernelParams kp;
    fa_bwd_params local;
    int rc = fill_bwd(user, kp, local, true);
    if (rc) return rc;
    const fa_bwd_params* p = &local;
    if (p->b == 0 || p->seqlen_k == 0) return FA_OK;
    return hip_status(fa::launch_bwd_dkdv(kp, p->dtype, (hipStream_t)stream), "fa_bwd_dkdv launch");
}

int fa_run_mha_bwd(const fa_bwd_params* user, void* stream) {
    fa::BwdKernelParams kp;
    fa_bwd_params local;
    int rc = fill_bwd(user, kp, local, true);
    if (rc) return rc;
    const fa_bwd_params* p = &local;
    if (p->b == 0) return FA_OK;
    hipStream_t s = (hipStream_t)stream;
    // The reference's run_flash_bwd launches dot_do_o, dQ, dK/dV (flash_bwd_launch_template.h:69-146).  Here the dQ kernel computes
    // D = rowsum(dO * O) for its own rows in its prologue and leaves it in dsoftmax_sum for the dK/dV launch: two launches, same
    // stream, no host sync in between.  (fa_bwd_dot_do_o stays available as a stand-alone entry point.)
    if (p->seqlen_q > 0) {
        if ((rc = hip_status(fa::launch_bwd_dq(kp, p->dtype, s), "fa_bwd_dq launch"))) return rc;
    }
    if (p->seqlen_k > 0) {
        if ((rc = hip_status(fa::launch_bwd_dkdv(kp, p->dtype, s), "fa_bwd_dkdv launch"))) return rc;
    }
    return FA_OK;
}

int fa_mha_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
               int32_t b, int32_t sq, int32_t sk, int32_t h, int32_t hk, int32_t d,
               int32_t dtype, int32_t is_causal, void* stream) {
    fa_fwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
    p.b = b; p.seqlen_q = sq; p.seqlen_k = sk; p.h = h; p.h_k = hk; p.d = d; p.dtype = dtype; p.is_causal = is_causal;
    p.q_stride = contiguous_bshd(sq, h, d); p.o_stride = p.q_stride;
    p.k_stride = contiguous_bshd(sk, hk, d); p.v_stride = p.k_stride;
    return fa_run_mha_fwd(&p, stream);
}

int fa_mha_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                      const int32_t* cu_q, const int32_t* cu_k,
                      int32_t b, int32_t max_sq, int32_t max_sk, int32_t h, int32_t hk, int32_t d,
                      int32_t dtype, int32_t is_causal, void* stream) {
    if (cu_q == nullptr || cu_k == nullptr) return fail(FA_ERR_NULL_POINTER, "cu_seqlens_q/cu_seqlens_k must not be NULL");
    fa_fwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.cu_seqlens_q = cu_q; p.cu_seqlens_k = cu_k;
    p.b = b; p.seqlen_q = max_sq; p.seqlen_k = max_sk; p.h = h; p.h_k = hk; p.d = d; p.dtype = dtype; p.is_causal = is_causal;
    p.q_stride = fa_strides{0, (int64_t)h * d, d}; p.o_stride = p.q_stride;
    p.k_stride = fa_strides{0, (int64_t)hk * d, d}; p.v_stride = p.k_stride;
    return fa_run_mha_fwd(&p, stream);
}

int fa_mha_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse,
               const void* dout, void* dq, void* dk, void* dv, float* dsum,
               int32_t b, int32_t sq, int32_t sk, int32_t h, int32_t hk, int32_t d,
               int32_t dtype, int32_t is_causal, void* stream) {
    fa_bwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.dout = dout; p.dq = dq; p.dk = dk; p.dv = dv; p.dsoftmax_sum = dsum;
    p.b = b; p.seqlen_q = sq; p.seqlen_k = sk; p.h = h; p.h_k = hk; p.d = d; p.dtype = dtype; p.is_causal = is_causal;
    p.q_stride = contiguous_bshd(sq, h, d); p.o_stride = p.do_stride = p.dq_stride = p.q_stride;
    p.k_stride = contiguous_bshd(sk, hk, d); p.v_stride = p.dk_stride = p.dv_stride = p.k_stride;
    return fa_run_mha_bwd(&p, stream);
}

int fa_mha_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse,
                      const void* dout, void* dq, void* dk, void* dv, float* dsum,
                      const int32_t* cu_q, const int32_t* cu_k,
                      int32_t b, int32_t max_sq, int32_t max_sk, int32_t h, int32_t hk, int32_t d,
                      int32_t dtype, int32_t is_causal, void* stream) {
    if (cu_q == nullptr || cu_k == nullptr) return fail(FA_ERR_NULL_POINTER, "cu_seqlens_q/cu_seqlens_k must not be NULL");
    fa_bwd_params p;
    FA_PARAMS_INIT(p);
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.dout = dout; p.dq = dq; p.dk = dk; p.dv = dv; p.dsoftmax_sum = dsum;
    p.cu_seqlens_q = cu_q; p.cu_seqlens_k = cu_k;
    p.b = b; p.seqlen_q = max_sq; p.seqlen_k = max_sk; p.h = h; p.h_k = hk; p.d = d; p.dtype = dtype; p.is_causal = is_causal;
    p.q_stride = fa_strides{0, (int64_t)h * d, d}; p.o_stride = p.do_stride = p.dq_stride = p.q_stride;
    p.k_stride = fa_strides{0, (int64_t)hk * d, d}; p.v_stride = p.dk_stride = p.dv_stride = p.k_stride;
    return fa_run_mha_bwd(&p, stream);
}

}  // extern "C"
