// fa_params.hpp — kernel-side parameter blocks (passed by value in the kernarg segment).
// Counterpart of the reference's Flash_fwd_params / Flash_bwd_params (src/flash.h:6-76), with
// real element strides and 64-bit offsets (the reference's int32 offsets in block_info.h:15-21
// overflow at b=32, s=16k, h=32, d=128 = 2^31 elements).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

struct TStride {
    int64_t batch, row, head;   // in elements
};

struct FwdKernelParams {
    const void* q_ptr;
    const void* k_ptr;
    const void* v_ptr;
    void* o_ptr;
    float* lse_ptr;
    const int32_t* cu_seqlens_q;
    const int32_t* cu_seqlens_k;
    TStride q, k, v, o;
    int64_t lse_row_stride;     // elements between consecutive (batch, head) rows of lse
    int32_t b, seqlen_q, seqlen_k, h, h_k, h_ratio, d;
    int32_t is_causal;
    uint32_t n_q_tiles;         // filled by the launcher
    uint32_t varlen_slots;      // filled by the launcher: != 0 -> compact varlen grid (fa_device.hpp), 0 -> plain grid
    uint32_t group_heads;       // filled by the launcher: dispatch order of a plain causal grid (fa_device.hpp:decode_block), 0 = one head after the other
    int64_t total_q;            // packed token count of q (0 = unknown)
    float scale_log2e;          // log2(e) / sqrt(d)
    float scale;                // 1 / sqrt(d)
};

struct BwdKernelParams {
    const void* q_ptr;
    const void* k_ptr;
    const void* v_ptr;
    const void* o_ptr;
    const void* do_ptr;
    const float* lse_ptr;
    float* dsum_ptr;            // D = rowsum(dO * O), layout of lse
    void* dq_ptr;
    void* dk_ptr;
    void* dv_ptr;
    const int32_t* cu_seqlens_q;
    const int32_t* cu_seqlens_k;
    TStride q, k, v, o, dout, dq, dk, dv;
    int64_t lse_row_stride;
    int32_t b, seqlen_q, seqlen_k, h, h_k, h_ratio, d;
    int32_t is_causal;
    uint32_t n_q_tiles, n_k_tiles;
    uint32_t varlen_slots;      // per launch, like FwdKernelParams
    uint32_t group_heads;       // per launch, like FwdKernelParams
    int64_t total_q, total_k;   // packed token counts (0 = unknown)
    float scale_log2e;
    float scale;
    // dK/dV with the query-head group of a KV head split over n_split workgroups (fa_bwd.hip): fp32 partial sums
    // ws[tensor 0 = dK, 1 = dV][split][key row of the whole batch][kv head][d]; n_split = 1: no workspace, direct output
    float* ws;
    int64_t ws_bytes;
    int64_t ws_rows;            // key rows of the whole batch: b * seqlen_k, or total_k for packed tensors
    int32_t n_split;
};

// Dispatch order of a plain causal grid (fa_device.hpp:decode_block): how many (batch, head) streams of an XCD are walked together, tile index first.
// Up to 4096 rows: all of them (longest-processing-time order across heads: -12..-30 % against one head after the other at 1k-4k).  Longer
// sequences: ~2 workgroups per compute unit at a time, ceil(2 x 32 x wg_per_cu / tiles) heads (2 heads at 8k rows and head_dim 128: -1..2 %; one
// head from 16k on = the order of rounds 1-3, where a head alone is two workgroups per unit and its K / V stays in the XCD's L2).  `wg_per_cu` =
// workgroups of this kernel a compute unit holds at a time (2 for the narrow head_dim-64 kernels); an XCD has 32 compute units.
// FA_CAUSAL_ORDER 0 = one head after the other everywhere (A/B).  profiles/r4_causal_tile_order_ab.log, r4_causal_tile_order_long_ab.log, r4_causal_group_order_ab.log
#ifndef FA_CAUSAL_ORDER
#define FA_CAUSAL_ORDER 1
#endif
inline uint32_t causal_group_heads(bool causal, bool compact_grid, int64_t n_bh, int64_t rows, int64_t tiles, int wg_per_cu) {
    if (FA_CAUSAL_ORDER == 0 || !causal || compact_grid || (n_bh & 7) != 0 || tiles < 2) return 0u;
    const int64_t per_xcd = n_bh >> 3;
    if (rows <= 4096) return (uint32_t)per_xcd;
    const int64_t g = (2 * 32 * (int64_t)wg_per_cu + tiles - 1) / tiles;
    return g <= 1 ? 0u : (uint32_t)(g < per_xcd ? g : per_xcd);
}

// query-head group split chosen for a dK/dV launch (1 = none) and the workspace it needs
int32_t dkdv_split(const BwdKernelParams& kp, int64_t avail_bytes);
int64_t dkdv_workspace_bytes(const BwdKernelParams& kp, int32_t n_split);

hipError_t launch_fwd(FwdKernelParams kp, int dtype, hipStream_t stream);
const char* fwd_kernel_name(int d);
const char* fwd_kernel_name_for(const FwdKernelParams& kp);                 // the kernel launch_fwd would pick for this problem
const char* bwd_kernel_name_for(const BwdKernelParams& kp, bool dkdv);      // ... launch_bwd_dq / launch_bwd_dkdv
int set_kernel_policy(int policy);      // FA_POLICY_* of the public header; returns the previous one, -1 for an unknown value
int kernel_policy();
hipError_t launch_bwd_dot_do_o(BwdKernelParams kp, int dtype, hipStream_t stream);
hipError_t launch_bwd_dq(BwdKernelParams kp, int dtype, hipStream_t stream);
hipError_t launch_bwd_dkdv(BwdKernelParams kp, int dtype, hipStream_t stream);

}  // namespace fa
