/*
 * flash_attn_gfx950.h — C ABI of the MI355X (gfx950 / CDNA4) fused attention library.
 *
 * This is the drop-in boundary for the hot path of ssiu/flash-attention-turing
 * (reference paths below are relative to the reference checkout):
 *
 *   reference entry (pybind, csrc/flash_attn/flash_api.cpp)      C-ABI replacement
 *   ---------------------------------------------------------   -----------------------------
 *   mha_fwd         flash_api.cpp:156-223  ("fwd",  :472)        fa_mha_fwd
 *   mha_bwd         flash_api.cpp:228-317  ("bwd",  :473)        fa_mha_bwd
 *   mha_varlen_fwd  flash_api.cpp:319-381  ("varlen_fwd", :474)  fa_mha_varlen_fwd
 *   mha_varlen_bwd  flash_api.cpp:383-468  ("varlen_bwd", :475)  fa_mha_varlen_bwd
 *   run_mha_fwd     flash_api.cpp:139-145  + Flash_fwd_params    fa_run_mha_fwd(fa_fwd_params)
 *   run_mha_bwd     flash_api.cpp:147-153  + Flash_bwd_params    fa_run_mha_bwd(fa_bwd_params)
 *                   (src/flash.h:6-76 are the param structs these mirror)
 *
 * Only plain pointers, integers and an opaque stream handle cross this boundary: no torch
 * types, no C++ types.  All pointers are DEVICE pointers (HBM) unless stated otherwise.
 * The library never allocates or frees device memory and never synchronises the device;
 * every kernel is enqueued on the `stream` argument (a hipStream_t passed as void*;
 * NULL = the legacy default stream, which is what the reference launches on,
 * flash_fwd_launch_template.h:68).
 *
 * Conventions (identical to the reference):
 *   q   : (batch, seqlen_q, nheads,   head_dim)   fp16 or bf16
 *   k,v : (batch, seqlen_k, nheads_k, head_dim)   same dtype, nheads % nheads_k == 0 (GQA/MQA)
 *   o   : like q, same dtype.   lse : (batch, nheads, seqlen_q) fp32, natural log.
 *   scale = 1/sqrt(head_dim) always (flash_fwd_kernel.h:351); causal mask is bottom-right
 *   aligned (mask.h:172): key j visible to query i iff j - i <= seqlen_k - seqlen_q.
 *   Rows with no visible key produce o = 0 and lse = 0.0 (flash_fwd_kernel.h:720-728,767-771).
 *   varlen: q (total_q, nheads, d), k/v (total_k, nheads_k, d) packed, cu_seqlens int32
 *   device arrays of length batch+1, lse padded to (batch, nheads, max_seqlen_q).
 *   head_dim in {64, 128} (static_switch.h:29-38); unlike the reference an unsupported
 *   head_dim is an error, not a silent no-op.
 *
 * Every function returns FA_OK (0) or a negative FA_ERR_* code; positive values are
 * hipError_t codes from the launch.  fa_last_error() gives a human readable message for
 * the calling thread.
 */
#ifndef FLASH_ATTN_GFX950_H
#define FLASH_ATTN_GFX950_H

#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_ABI_VERSION 4   /* 2: optional total_q / total_k appended to fa_fwd_params / fa_bwd_params; 3: optional workspace appended to
                              fa_bwd_params; 4: both structs start with {struct_size, magic}, appended fields are read only if the caller's
                              struct holds them */
/* Every params struct starts with its own size AS THE CALLER COMPILED IT and this constant.  The library copies min(struct_size,
 * its own sizeof) bytes into a zeroed local struct: a caller built against an older ABI >= 4 header simply leaves the fields appended
 * since then at their defaults (0 / NULL = "not given"), and a caller built against an ABI 1-3 header (no size field: its first 8
 * bytes are the q pointer, whose upper half can never equal the magic) is rejected with FA_ERR_BAD_ABI instead of having the library
 * read past the end of its struct.  FA_PARAMS_INIT(p) zeroes a struct and fills both fields. */
#define FA_PARAMS_MAGIC 0xFA950A71u
#define FA_PARAMS_INIT(p) do { memset(&(p), 0, sizeof(p)); (p).struct_size = (uint32_t)sizeof(p); (p).magic = FA_PARAMS_MAGIC; } while (0)

enum fa_dtype { FA_FP16 = 0, FA_BF16 = 1 };

enum fa_status {
    FA_OK = 0,
    FA_ERR_NULL_POINTER = -1,
    FA_ERR_BAD_SHAPE = -2,       /* rank/size disagreement (reference: TORCH_CHECKs flash_api.cpp:178-183) */
    FA_ERR_BAD_GQA = -3,         /* nheads % nheads_k != 0 (flash_api.cpp:183) */
    FA_ERR_BAD_HEADDIM = -4,     /* head_dim not in {64,128} */
    FA_ERR_BAD_DTYPE = -5,
    FA_ERR_BAD_STRIDE = -6,      /* last dim not contiguous, misaligned rows, or extent > 2^31 bytes per sequence */
    FA_ERR_NO_DEVICE = -7,
    FA_ERR_BAD_ABI = -8          /* params struct without a valid {struct_size, magic} header, shorter than the ABI 4 base, or longer
                                    than this library knows (caller compiled against a newer header) */
};

/* Element strides (NOT bytes). The innermost (head_dim) stride is 1 by contract. */
typedef struct fa_strides {
    int64_t batch; /* ignored for varlen (packed) tensors */
    int64_t row;   /* between consecutive sequence positions */
    int64_t head;  /* between consecutive heads */
} fa_strides;

/* Mirrors Qkv_params + Flash_fwd_params (reference src/flash.h:6-52). */
typedef struct fa_fwd_params {
    uint32_t struct_size;       /* sizeof(fa_fwd_params) in the caller's translation unit */
    uint32_t magic;             /* FA_PARAMS_MAGIC */
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;                 /* (b, h, seqlen_q) contiguous; varlen: (b, h, max_seqlen_q) */
    const int32_t* cu_seqlens_q; /* NULL for fixed-length batches */
    const int32_t* cu_seqlens_k;
    int32_t b;
    int32_t seqlen_q;           /* varlen: max_seqlen_q */
    int32_t seqlen_k;           /* varlen: max_seqlen_k */
    int32_t h;
    int32_t h_k;
    int32_t d;
    int32_t dtype;              /* enum fa_dtype */
    int32_t is_causal;
    fa_strides q_stride, k_stride, v_stride, o_stride;
    /* ABI 2, optional (0 = unknown).  varlen only: number of rows of the PACKED q / k tensors (what the reference's mha_varlen_fwd
     * sees as q.size(0) / k.size(0), flash_api.cpp:319-381).  MUST be >= cu_seqlens_q[b] / cu_seqlens_k[b] (like max_seqlen, the
     * library cannot check device values without synchronising).  When given, the launch grid is sized by the tokens actually
     * present instead of max_seqlen x batch, which matters for batches of very unequal lengths (varlen). */
    int64_t total_q;
    int64_t total_k;
} fa_fwd_params;

/* Mirrors Flash_bwd_params (reference src/flash.h:55-76). */
typedef struct fa_bwd_params {
    uint32_t struct_size;       /* sizeof(fa_bwd_params) in the caller's translation unit */
    uint32_t magic;             /* FA_PARAMS_MAGIC */
    const void* q;
    const void* k;
    const void* v;
    const void* o;
    const void* dout;
    const float* lse;
    void* dq;
    void* dk;                   /* (b, seqlen_k, h_k, d): already summed over the GQA group */
    void* dv;
    float* dsoftmax_sum;        /* D = rowsum(dO*O), same shape as lse (the reference's do_o): written by the dQ launch, read by dK/dV */
    const int32_t* cu_seqlens_q;
    const int32_t* cu_seqlens_k;
    int32_t b;
    int32_t seqlen_q;
    int32_t seqlen_k;
    int32_t h;
    int32_t h_k;
    int32_t d;
    int32_t dtype;
    int32_t is_causal;
    fa_strides q_stride, k_stride, v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride;
    int64_t total_q;            /* ABI 2, optional, see fa_fwd_params */
    int64_t total_k;
    /* ABI 3, optional (NULL / 0 = none): fp32 scratch for the dK/dV kernel.  Its grid is b * h_k * ceil(seqlen_k / 128)
     * workgroups; with few KV heads (GQA / MQA) that underfills the chip and, under a causal mask, is unbalanced (the first key
     * block of a sequence meets every query tile, the last a single one).  Given this scratch, the h / h_k query heads of a KV
     * head are dealt to several workgroups that leave fp32 partial sums here, and a second kernel adds them in a fixed order
     * (results stay deterministic).  fa_bwd_workspace_bytes() is the size that lets the library choose freely; a smaller buffer
     * limits the split, none keeps the single-pass behaviour of ABI 2.  Contents are unspecified afterwards. */
    void* workspace;
    int64_t workspace_bytes;
} fa_bwd_params;

/* ---- library info ---------------------------------------------------------------------- */
int fa_abi_version(void);
const char* fa_last_error(void);
const char* fa_build_info(void);          /* e.g. "gfx950 hipcc ..." */

/* ---- param-struct entry points (replace run_mha_fwd / run_mha_bwd) ---------------------- */
int fa_run_mha_fwd(const fa_fwd_params* params, void* stream);
int fa_run_mha_bwd(const fa_bwd_params* params, void* stream);

/* ---- flat entry points for contiguous tensors (replace mha_fwd / mha_bwd / varlen_*) ---- */
/* q (b,sq,h,d), k/v (b,sk,hk,d), o (b,sq,h,d), lse (b,h,sq); all contiguous. */
int fa_mha_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
               int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t h_k, int32_t d,
               int32_t dtype, int32_t is_causal, void* stream);

int fa_mha_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse,
               const void* dout, void* dq, void* dk, void* dv, float* dsoftmax_sum,
               int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t h_k, int32_t d,
               int32_t dtype, int32_t is_causal, void* stream);

/* q (total_q,h,d), k/v (total_k,hk,d) packed; lse (b,h,max_seqlen_q). */
int fa_mha_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                      const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                      int32_t b, int32_t max_seqlen_q, int32_t max_seqlen_k,
                      int32_t h, int32_t h_k, int32_t d,
                      int32_t dtype, int32_t is_causal, void* stream);

int fa_mha_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse,
                      const void* dout, void* dq, void* dk, void* dv, float* dsoftmax_sum,
                      const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                      int32_t b, int32_t max_seqlen_q, int32_t max_seqlen_k,
                      int32_t h, int32_t h_k, int32_t d,
                      int32_t dtype, int32_t is_causal, void* stream);

/* D = rowsum(dO * O) alone (replaces flash_bwd_dot_do_o_kernel, flash_bwd_preprocess_kernel.h:23-96): the one HBM-bound kernel of
 * the reference's path.  Since round 2 fa_run_mha_bwd no longer launches it (the dQ kernel computes D in its prologue); it stays as a
 * stand-alone entry point and is measured separately. */
int fa_bwd_dot_do_o(const fa_bwd_params* params, void* stream);
/* The other two launches of run_flash_bwd, individually (flash_bwd_launch_template.h:95-146): dQ and dK/dV.  fa_bwd_dq computes D for
 * its rows itself (from O and dO) and WRITES params->dsoftmax_sum; fa_bwd_dkdv READS it, i.e. fa_bwd_dq (or fa_bwd_dot_do_o) must
 * have run on the same stream before.  fa_run_mha_bwd == fa_bwd_dq then fa_bwd_dkdv; exposed so that each kernel can be timed /
 * profiled against its own roofline (bench.py `roofline_bwd`). */
int fa_bwd_dq(const fa_bwd_params* params, void* stream);
int fa_bwd_dkdv(const fa_bwd_params* params, void* stream);
/* Bytes of fa_bwd_params.workspace the dK/dV launch of these params would use (0: it would not split).  The `workspace` /
 * `workspace_bytes` fields of the argument are ignored (a stale or unaligned pointer left in a reused struct is not an error here).
 * Host-only arithmetic; the split target follows the CU count of the current device (256 when there is none).  Negative = error code. */
int64_t fa_bwd_workspace_bytes(const fa_bwd_params* params);

/* ---- measurement helpers ----------------------------------------------------------------- */
/* Algorithmic FLOPs of one forward call (4*b*h*sq*sk*d, causal counts only visible pairs);
 * backward = 2.5x this. Host-only arithmetic, no GPU needed. */
double fa_fwd_flops(int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t d, int32_t is_causal);
/* Algorithmic HBM bytes of one forward call: q,k,v,o once + lse. */
double fa_fwd_bytes(int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t h_k, int32_t d);
/* Name of the forward kernel the library dispatches LARGE problems of this head_dim to (what a profiler's kernel trace of the
 * BASELINE configurations will show; lets a benchmark tie a committed PMC profile to the kernel that actually ran).  head_dim 128
 * has two kernels: fa_fwd_pp16_kernel serves problems of seqlen_q * seqlen_k >= 2^20 whose launch fills the chip (fa_set_kernel_policy below), fa_fwd_pp_kernel
 * the others; head_dim 64 likewise for fp16 inputs from 2^24 (2^26 under a causal mask) - the answer given here - while bf16
 * inputs stay on fa_fwd_pp_kernel at every size (fa_kernel_name_dtype answers per dtype). */
const char* fa_fwd_kernel_name(int32_t d);
/* head_dim 128 has two sets of kernels, tiled for v_mfma_f32_32x32x16 and for v_mfma_f32_16x16x32.  Both meet the same tolerances;
 * they differ in speed only: the 16x16x32 shape draws less power per FLOP and wins where the chip's power cap binds (long launches),
 * the 32x32x16 forward needs fewer cycles and wins short ones and launches that leave compute units idle.  FA_POLICY_AUTO (the default, head_dim 128, round 6): the
 * forward and dQ go to the 16x16x32 set when the launch has at least one 256-row workgroup per compute unit (batch x heads x ceil(seqlen_q / 256) >= CUs; under a causal
 * mask two per unit, or one with seqlen_q * seqlen_k >= 2^26) and, forward and causal dQ, seqlen_q * seqlen_k >= 2^20; dK/dV from seqlen_q * seqlen_k >= 2^20 whatever the
 * launch (from 2^18 when it brings two 128-key workgroups per compute unit).  Because the choice follows the LAUNCH, a (batch, head) shard of a problem may get the other kernel set than the whole problem would: a caller that wants the
 * whole problem's kernels, hence its bits, on every shard states the whole problem's batch x heads with fa_set_policy_problem_heads() before it runs the shards (rounds
 * 3-5 chose per head only; flash_attn_turing/sharding.py:problem_policy does it for the Python surface).  One more exception stays: dK / dV of a GQA / MQA call that is given a
 * workspace - how far a head group is split, hence the order its partial sums are added in, follows the launch's workgroup count and the
 * device's CU count; shards then agree with the whole problem to a last rounding, not bit for bit.  FA_POLICY_MFMA32 / FA_POLICY_MFMA16 pin one set for every launch.  Process-wide, thread-safe; returns the
 * previous policy, -1 (and changes nothing) for an unknown value.  head_dim 64 has both forward kernels since round 4 and both backward sets since round 5; there the
 * launch matters the other way round: the 32x32x16 kernels live off two co-resident workgroups per compute unit, so FA_POLICY_AUTO keeps them for launches that fill the
 * chip with short sequences and gives the 16x16x32 set (one workgroup per unit) the long sequences - fp16 forward from 2^24 pairs per head (2^26 causal), dQ without a mask
 * from 2^18 and under one from 2^26, dK/dV from 2^24 (2^28 causal) - AND, since round 6, every launch that leaves the second workgroup slot empty (forward: at most one
 * 256-row workgroup per unit from 2^22 pairs, two under a mask, half a one from 2^20; dQ / dK/dV under a mask: at most two per unit; 5-25 % there), never when a causal
 * problem has fewer keys than queries; bf16 forwards stay on the 32x32x16 kernel (-8..-11 % / -4..-6 %
 * where it serves; both dtypes).  The pinned policies apply to every head_dim, stage and dtype.  The reference has no counterpart.
 * Precision contract of the forward's softmax row sums: fp16 inputs on the 16x16x32 kernel sum the ROUNDED P in the matrix pipe after an exactly
 * summed prefix of 1024 keys (LSE within 5e-5 of fp32 math at the BASELINE sizes); bf16 inputs keep exact fp32 VALU sums on every kernel (LSE within
 * 2e-6): with bf16's 8-bit P the same construction measured 1.2e-4 even behind a 4096-key exact prefix, for -0.9 % at configs[3] - refused
 * (profiles/r5_fwd_bf16_mfma_rowsum_ab.log, r5_lse_error_bf16_mfma_rowsum.json). */
#define FA_POLICY_MFMA32 0
#define FA_POLICY_MFMA16 1
#define FA_POLICY_AUTO 2
int32_t fa_set_kernel_policy(int32_t policy);
/* The batch x heads FA_POLICY_AUTO sizes every following launch with, instead of the launch's own: the WHOLE problem's, stated by a caller that runs (batch, head) shards of
 * it one after the other or on several devices and wants every shard served by the kernels the whole problem would get (bit-identical results).  0 (the initial value) = the
 * launch's own batch x heads.  Process-wide, thread-safe; returns the previous value, -1 (and changes nothing) for a negative argument.  Round 6; the reference has no counterpart. */
int64_t fa_set_policy_problem_heads(int64_t batch_times_heads);
/* Name of the kernel a launch of this shape is dispatched to under the current policy (what a profiler's kernel trace will show):
 * stage FA_STAGE_FWD / FA_STAGE_DQ / FA_STAGE_DKDV; seqlen_* = the max_seqlen_* of a packed call.  "" for an unknown stage. */
#define FA_STAGE_FWD 0
#define FA_STAGE_DQ 1
#define FA_STAGE_DKDV 2
const char* fa_kernel_name(int32_t stage, int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t d, int32_t is_causal);
/* The same per input dtype (enum fa_dtype; fa_kernel_name answers for FA_FP16): the only choice that depends on it is the head_dim-64 forward
 * under FA_POLICY_AUTO.  "" for an unknown stage or dtype. */
const char* fa_kernel_name_dtype(int32_t stage, int32_t dtype, int32_t b, int32_t seqlen_q, int32_t seqlen_k, int32_t h, int32_t d, int32_t is_causal);
/* Peak shader clock of `device` in kHz (hipDeviceAttributeClockRate), or a negative HIP error code: with 256 CUs x 4096 FLOP/clk/CU
 * it derives the dense fp16 MFMA peak a benchmark quotes (256 x 2.4 GHz x 4096 = 2.5 PFLOP/s). */
int fa_device_clock_khz(int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* FLASH_ATTN_GFX950_H */
