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
// Only where the mask makes the tiles UNEVEN: seqlen_k < 2 * seqlen_q (with many more keys than queries - chunked prefill, sq 1k against sk 8k -
// every query tile sees nearly all keys, there is nothing to balance, and walking the heads together would stream each head's K / V once per
// tile instead of once: measured +23 %).  Up to 4096 rows and keys: all heads of the XCD together (longest-processing-time order across heads:
// -12..-30 % against one head after the other at 1k-4k).  Longer: ~2 workgroups per compute unit at a time, ceil(2 x 32 x wg_per_cu / tiles) heads (2
// heads at 8k rows and head_dim 128: -1..2 %; one head from 16k on = the order of rounds 1-3, where a head alone is two workgroups per unit and
// its K / V stays in the XCD's L2).  `tiles` / `rows` are those of the axis the grid walks (query tiles; key blocks for dK/dV), `wg_per_cu` the
// workgroups of this kernel a compute unit holds at a time (2 for the narrow head_dim-64 kernels); an XCD has 32 compute units.
// FA_CAUSAL_ORDER 0 = one head after the other everywhere (A/B).  profiles/r4_causal_tile_order_ab.log, r4_causal_tile_order_long_ab.log, r4_causal_group_order_ab.log
#ifndef FA_BWD_DMA_SAVE_M0
#define FA_BWD_DMA_SAVE_M0 0  // backward kernels: 1 = save / restore M0 around every hand-issued LDS-DMA piece (fa_device.hpp:dma16_to_lds_hidden).  Like the forward
                              // kernels they hold no compiler-visible LDS-DMA and nothing hipcc generates for them touches M0 (tests/test_kernel_resources_cpu.py keeps
                              // that true), so the two SALU per piece are dropped: -0.1..-1.4 %, bit-identical (profiles/r4_bwd_dma_no_m0_save_ab.log)
#endif
#ifndef FA_CAUSAL_ORDER
#define FA_CAUSAL_ORDER 1
#endif
// Packed batches on the compact grid (`compact_b` = number of sequences, 0 = plain grid): heaviest items first across sequences and heads
// (fa_device.hpp:varlen_slot_lookup_heavy_first) when there is one sequence per lane and at most 64 tiles per sequence.
#ifndef FA_VARLEN_HEAVY_FIRST
#define FA_VARLEN_HEAVY_FIRST 1
#endif
// FA_CAUSAL_GROUP_MB: up to 4096 rows the heads of an XCD are walked together only as far as the streams they re-read (K and V of a head for the
// forward / dQ, Q and dO for dK/dV) stay in reach of the cache behind the L2s: 16 x 32 heads at 2k rows in ONE group stream 4 GB per launch and
// run at 0.91 of the head-after-head time where groups of 12-24 MiB reach 0.79-0.80 (profiles/r4_causal_group_footprint_ab.log).
#ifndef FA_CAUSAL_GROUP_MB
#define FA_CAUSAL_GROUP_MB 16
#endif
inline uint32_t causal_group_heads(bool causal, int64_t compact_b, int64_t n_bh, int64_t seqlen_q, int64_t seqlen_k, int64_t tiles, int wg_per_cu,
                                   int64_t stream_bytes_per_head = 0) {
    if (FA_CAUSAL_ORDER == 0 || !causal || tiles < 2 || seqlen_k >= 2 * seqlen_q) return 0u;
    if (compact_b != 0)      // (n_bh = heads of the grid here: the XCD rule needs a multiple of 8)
        return (FA_VARLEN_HEAVY_FIRST && compact_b <= 64 && tiles <= 64 && (n_bh & 7) == 0) ? 0xffffffffu : 0u;
    if ((n_bh & 7) != 0) return 0u;
    const int64_t per_xcd = n_bh >> 3;
    int64_t g = (2 * 32 * (int64_t)wg_per_cu + tiles - 1) / tiles;                  // ~2 workgroups per compute unit
    if (seqlen_q <= 4096 && seqlen_k <= 4096) {
        // all heads together, as far as their streams fit; groups of equal size (a short last group runs unbalanced)
        int64_t cap = stream_bytes_per_head > 0 ? ((int64_t)FA_CAUSAL_GROUP_MB << 20) / stream_bytes_per_head : per_xcd;
        if (cap < g) cap = g;
        if (cap >= per_xcd) return (uint32_t)per_xcd;
        // equal groups (a short last group runs unbalanced: 8 % in profiles/r4_causal_group_order_ab.log): the largest divisor of the heads per
        // XCD within the cap, unless that is less than half of it - then near-equal groups of ceil(heads / groups)
        int64_t div = 1;
        for (int64_t c = cap; c >= 1; --c)
            if (per_xcd % c == 0) { div = c; break; }
        const int64_t n_groups = (per_xcd + cap - 1) / cap;
        g = 2 * div >= cap ? div : (per_xcd + n_groups - 1) / n_groups;
    }
    return g <= 1 ? 0u : (uint32_t)(g < per_xcd ? g : per_xcd);
}

// query-head group split chosen for a dK/dV launch (1 = none) and the workspace it needs
int32_t dkdv_split(const BwdKernelParams& kp, int64_t avail_bytes);
int64_t dkdv_workspace_bytes(const BwdKernelParams& kp, int32_t n_split);

hipError_t launch_fwd(FwdKernelParams kp, int dtype, hipStream_t stream);
const char* fwd_kernel_name(int d);
const char* fwd_kernel_name_for(const FwdKernelParams& kp, int dtype);                 // the kernel launch_fwd would pick for this problem
const char* bwd_kernel_name_for(const BwdKernelParams& kp, bool dkdv);      // ... launch_bwd_dq / launch_bwd_dkdv
int set_kernel_policy(int policy);      // FA_POLICY_* of the public header; returns the previous one, -1 for an unknown value
int kernel_policy();
// FA_POLICY_AUTO sizes a head_dim-128 launch by its workgroup count (round 6): (batch x heads) x tiles.  A caller that runs a (batch, head) SHARD of a problem and wants the
// kernels - hence the bits - of the whole problem states the whole problem's batch x heads here (0 = the launch's own); flash_attn_turing/sharding.py:problem_policy
int64_t set_policy_problem_heads(int64_t batch_times_heads);      // returns the previous value, -1 (and changes nothing) for a negative one
int64_t policy_bh(int64_t b, int64_t h);                          // the batch x heads FA_POLICY_AUTO sizes this launch with
int64_t device_cu_count();                                         // compute units of the current device (256 without one)
hipError_t launch_bwd_dot_do_o(BwdKernelParams kp, int dtype, hipStream_t stream);
hipError_t launch_bwd_dq(BwdKernelParams kp, int dtype, hipStream_t stream);
hipError_t launch_bwd_dkdv(BwdKernelParams kp, int dtype, hipStream_t stream);

}  // namespace fa
