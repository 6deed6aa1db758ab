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
    uint32_t tile_major;        // filled by the launcher: plain grid walked tile index first (fa_device.hpp:decode_block)
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
    uint32_t tile_major;        // per launch, like FwdKernelParams
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

// Plain causal grids of sequences up to this many rows are walked tile index first (fa_device.hpp:decode_block): longest-processing-time
// order while a compute unit sees only a few workgroups.  Measured (profiles/r4_causal_tile_order_ab.log, r4_causal_tile_order_long_ab.log):
// forward / dQ -12..-30 % at 1k-4k, -2..-9 % at 8k, dK/dV -3..-25 % up to 4k but +3.6 % at 8k (head_dim 128); at 16k every kernel loses
// 6-12 % (a head's K / V no longer stays in its XCD's L2 while its tiles run at different times).  0 = never (A/B switch).
#ifndef FA_TILE_MAJOR_MAX_ROWS
#define FA_TILE_MAJOR_MAX_ROWS 8192
#endif
#ifndef FA_TILE_MAJOR_MAX_ROWS_DKDV
#define FA_TILE_MAJOR_MAX_ROWS_DKDV (FA_TILE_MAJOR_MAX_ROWS < 4096 ? FA_TILE_MAJOR_MAX_ROWS : 4096)
#endif
inline uint32_t tile_major_for(bool causal, bool compact_grid, int64_t n_bh, int64_t rows, int64_t tiles, int64_t max_rows = FA_TILE_MAJOR_MAX_ROWS) {
    return (max_rows > 0 && causal && !compact_grid && (n_bh & 7) == 0 && tiles >= 2 && rows <= max_rows) ? 1u : 0u;
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
