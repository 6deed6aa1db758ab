// fa_bwd.hip — fused attention backward for MI355X (gfx950, CDNA4).
//
// The kernels of the reference's run_flash_bwd (csrc/flash_attn/src/flash_bwd_launch_template.h:69-146), re-designed for
// wave64 / MFMA:
//
//   fa_bwd_dot_do_o_kernel  D[b,h,i] = sum_d dO*O                 (flash_bwd_preprocess_kernel.h:23-96)
//   fa_bwd_dq_kernel        dQ = scale * sum_j dS_ij K_j           (flash_bwd_kernel.h:28-825)
//   fa_bwd_dkdv_kernel      dV = sum_i P_ij^T dO_i, dK = scale * sum_i dS_ij^T Q_i   (:842-1676)
//   fa_bwd_sum_splits_kernel  adds the fp32 partial dK / dV planes when the dK/dV launch split a GQA head group (C ABI 3)
//
// (head_dim 128 has a second dQ and a second dK/dV kernel, tiled for v_mfma_f32_16x16x32: fa_bwd_dq16.hip, fa_bwd_dkdv16.hip; the launchers at
// the end of this file pick per launch.)
//
// A backward call is TWO launches on one stream, dQ then dK/dV (plus the plane sum when split): the dQ kernel computes D for
// its own rows in its prologue and leaves it in dsoftmax_sum for dK/dV; the stand-alone dot_do_o kernel stays as an entry point.
// ("workspace" below always means the optional fp32 dK / dV scratch of C ABI 3, never D.)
//
// Like the reference's alternative this is the deterministic, atomics-free 7-GEMM form (S and dP are recomputed in both kernels); fp32
// atomics run at 1.33 TB/s on this chip, the single-kernel form would spend 26 ms accumulating dQ at C4 (profiles/r3_atomicbench.log).  Differences that matter on CDNA4:
//   * dQ kernel uses the forward's "swapped" layout (lane = query column), so LSE_i and D_i
//     are lane scalars and dS^T feeds the dQ^T MFMA straight from registers;
//   * dK/dV kernel uses the un-swapped layout (lane = key column), P and dS feed the dV^T / dK^T
//     MFMAs straight from registers, Q^T / dO^T operands come from hardware transposing LDS reads,
//     the dK^T / dV^T accumulators live in AGPRs;
//   * the GQA group loop is fused into the dK/dV kernel (the reference materialises per-q-head
//     dK/dV and reduces with torch::sum_out, flash_api.cpp:265-272,301-312); with caller-provided fp32 scratch the group is
//     split over workgroups when the grid would otherwise be small or causally unbalanced;
//   * operands that never change inside a kernel's loop live in registers (Q / dO fragments in dQ, K and part of V in dK/dV).
#include <atomic>
#include <type_traits>
#include "fa_bwd_dkdv_common.hpp"

namespace fa {

// =============================================================================================
// D = rowsum(dO * O): HBM-bound streaming kernel. 16 lanes x 16 B cover one 128-wide row
// (8 lanes for d = 64); 4 shuffle steps reduce within the lane group.
// =============================================================================================
constexpr int kDotThreads = 256;
constexpr int kDotRowsPerBlock = 64;

template <typename T, int D>
__global__ __launch_bounds__(kDotThreads) void fa_bwd_dot_do_o_kernel(const BwdKernelParams p) {
    constexpr int LPR = D / 8;                       // lanes per row
    constexpr int RPI = kDotThreads / LPR;           // rows per iteration
    int batch = blockIdx.z, mblk = blockIdx.x;
    const int head = blockIdx.y;
    if (p.varlen_slots != 0) {       // compact varlen grid: blockIdx.x = slot, blockIdx.z = 0
        int nt;
        if (!varlen_slot_lookup<kDotRowsPerBlock>(p.cu_seqlens_q, p.b, blockIdx.x, batch, mblk, nt)) return;
    }
    int sq = p.seqlen_q;
    int64_t row0 = 0, o_boff = (int64_t)batch * p.o.batch, do_boff = (int64_t)batch * p.dout.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int beg = p.cu_seqlens_q[batch];
        sq = min(p.cu_seqlens_q[batch + 1] - beg, p.seqlen_q);   // D rows are padded to max_seqlen_q: never write past them
        row0 = beg;
        o_boff = do_boff = 0;
    }
    const int m0 = mblk * kDotRowsPerBlock;
    if (m0 >= sq) return;
    const T* o_base = (const T*)p.o_ptr + o_boff + row0 * p.o.row + (int64_t)head * p.o.head;
    const T* do_base = (const T*)p.do_ptr + do_boff + row0 * p.dout.row + (int64_t)head * p.dout.head;
    float* d_base = p.dsum_ptr + ((int64_t)batch * p.h + head) * p.lse_row_stride;
    const int sub = threadIdx.x % LPR, rlocal = threadIdx.x / LPR;
#pragma unroll
    for (int it = 0; it < kDotRowsPerBlock / RPI; ++it) {
        const int row = m0 + it * RPI + rlocal;
        float acc = 0.f;
        if (row < sq) {
            const u32x4 a = *(const u32x4*)(o_base + (int64_t)row * p.o.row + sub * 8);
            const u32x4 b = *(const u32x4*)(do_base + (int64_t)row * p.dout.row + sub * 8);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                acc += LP<T>::to_float((uint16_t)(a[w] & 0xffff)) * LP<T>::to_float((uint16_t)(b[w] & 0xffff));
                acc += LP<T>::to_float((uint16_t)(a[w] >> 16)) * LP<T>::to_float((uint16_t)(b[w] >> 16));
            }
        }
#pragma unroll
        for (int s = LPR / 2; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
        if (sub == 0 && row < sq) d_base[row] = acc;
    }
}

// =============================================================================================
// dQ kernel: workgroup = 8 waves = 256 query rows of one (batch, head); loops over 64-key tiles.
// =============================================================================================
constexpr int kDqThreads = 512;
constexpr int kDqBlockM = 256;
constexpr int kDqBlockN = 64;

template <typename T, int D, bool CAUSAL>
// D = 64: ask for 4 waves per SIMD = TWO 8-wave workgroups per CU in dQ and dK/dV (blocks come in units of 2 waves per
// SIMD; with the default bound the compiler took 171 / 88+64 registers and one workgroup per CU).  At 128 registers dQ
// non-causal carries 1 scratch op per tile and dK/dV 2-3 (tests/test_kernel_resources_cpu.py allows exactly that, nothing
// else).  Interleaved A/B of the whole backward, outputs bit-identical (profiles/r1_bwd_d64_occupancy_ab.log): dK/dV alone
// 0.80-0.90x time, dQ alone 0.90-0.98x, both 0.78-0.83x on every D = 64 shape (8k / 2k / 512, causal or not, GQA); D = 128
// unchanged (its registers and LDS allow one workgroup per CU only).
#define FA_DQ_MIN_WAVES(D) ((D) == 64 ? 4 : 2)
#ifndef FA_DQ_STAGGER_DMA
#define FA_DQ_STAGGER_DMA 1
#endif
#ifndef FA_DQ_NO_UNROLL
#define FA_DQ_NO_UNROLL 0
#endif
// dK/dV: k-steps whose K / V fragments are held in registers (D = 128: all 8 of K, 2 of V - the most that fits the 128 VGPRs next to
// the 128 accumulator registers without a spill; all 8 of V spills and runs 1.4-1.8x slower; D = 64 runs on 64 + 64 registers: none).
// Timing-only ablation first (profiles/r2_bwd_dkdv_lds_ab.log): no K / V fragment reads at all = -14..-18 % of the kernel.
#ifndef FA_KV_KREG
#define FA_KV_KREG(D) ((D) == 128 ? 8 : 0)
#endif
#ifndef FA_KV_VREG
#define FA_KV_VREG(D) ((D) == 128 ? 2 : 0)
#endif
#define FA_KV_MIN_WAVES(D) ((D) == 64 ? 4 : 2)
// dK/dV: when the Q / dO tile DMA is issued.  Per-phase s_memtime stamps (tools/phase_timing_dkdv.py, profiles/r3_dkdv_phase_timing.log):
// an LDS-DMA piece costs the issuing wave ~110 cycles, and with every wave issuing its pieces at the top of the iteration both waves
// of a SIMD are in that phase together - the matrix pipe idles through it.  1: waves 4-7 issue theirs AFTER their S / dP MFMAs, while
// waves 0-3 (which issue at the top) are still in that phase: -0..1 % at D = 128, -10 % at D = 64 causal (profiles/r3_dkdv_ab.log).
// Measured and removed in round 3 (same log): waves 0-3 issuing ALL pieces at the end of their tile in the ~1000 cycles they wait
// at the barrier (3-deep rings through the K tile's region: +-0 non-causal, +3..5 % causal), and a two-group ping-pong schedule of
// the whole kernel (one phase apart, 3-deep rings, hand-pipelined phases: +3..39 % - a wave ALONE on the matrix pipe is latency-
// bound on its LDS fragments, the lock-step pair hides each other's round trips, and 128 + 128 registers leave no room to
// prefetch deeper; history: commit 9a378a6).
#ifndef FA_KV_PF2
#define FA_KV_PF2(D) ((D) == 64 ? 2 : 3)      // dV / dK phase: transposed fragments in flight (D = 64: 3 costs 2 spill ops per tile in its 64 + 64
#endif                                        // registers; 4 at D = 128 measured +-1 %, profiles/r3_dkdv_ab.log)
#ifndef FA_KV_STAGGER_DMA
#define FA_KV_STAGGER_DMA 1
#endif
__global__ __launch_bounds__(kDqThreads, FA_DQ_MIN_WAVES(D)) void fa_bwd_dq_kernel(const BwdKernelParams p) {
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int TILEB = kDqBlockN * ROWB;
    __shared__ __attribute__((aligned(16))) char smem_raw[(4 * TILEB > kDqBlockM * ROWB) ? 4 * TILEB : kDqBlockM * ROWB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tile, batch, head, tiles_seq;
    if (!decode_work<kDqBlockM>(blockIdx.x, p.n_q_tiles, p.varlen_slots, p.cu_seqlens_q, p.b, p.h, tile, batch, head, tiles_seq, p.group_heads)) return;
    if (CAUSAL) tile = tiles_seq - 1 - tile;
    const int head_k = head / p.h_ratio;

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch, v_boff = (int64_t)batch * p.v.batch,
            do_boff = (int64_t)batch * p.dout.batch, dq_boff = (int64_t)batch * p.dq.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int qb = p.cu_seqlens_q[batch], kb = p.cu_seqlens_k[batch];
        sq = min(p.cu_seqlens_q[batch + 1] - qb, p.seqlen_q);   // clamp to the declared max_seqlen_q (padded LSE / D rows)
        sk = p.cu_seqlens_k[batch + 1] - kb;
        q_row0 = qb; k_row0 = kb;
        q_boff = k_boff = v_boff = do_boff = dq_boff = 0;
    }
    const int m0 = tile * kDqBlockM;
    if (m0 >= sq) return;
    const int delta = sk - sq;
    const int rows_here = min(kDqBlockM, sq - m0);

    const T* q_base = uniform_ptr((const T*)p.q_ptr + q_boff + (q_row0 + m0) * p.q.row + (int64_t)head * p.q.head);
    const T* do_base = uniform_ptr((const T*)p.do_ptr + do_boff + (q_row0 + m0) * p.dout.row + (int64_t)head * p.dout.head);
    const T* o_base = uniform_ptr((const T*)p.o_ptr + (p.cu_seqlens_q != nullptr ? 0 : (int64_t)batch * p.o.batch) + (q_row0 + m0) * p.o.row + (int64_t)head * p.o.head);
    T* dq_base = uniform_ptr((T*)p.dq_ptr + dq_boff + (q_row0 + m0) * p.dq.row + (int64_t)head * p.dq.head);
    const T* k_base = uniform_ptr((const T*)p.k_ptr + k_boff + k_row0 * p.k.row + (int64_t)head_k * p.k.head);
    const T* v_base = uniform_ptr((const T*)p.v_ptr + v_boff + k_row0 * p.v.row + (int64_t)head_k * p.v.head);
    const int64_t stat_off = ((int64_t)batch * p.h + head) * p.lse_row_stride + m0;

    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), do_rowb = (uint32_t)(p.dout.row * 2), dq_rowb = (uint32_t)(p.dq.row * 2),
                   k_rowb = (uint32_t)(p.k.row * 2), v_rowb = (uint32_t)(p.v.row * 2);
    const rsrc_t q_rs = make_rsrc(q_base, (uint32_t)(rows_here - 1) * q_rowb + ROWB);
    const rsrc_t do_rs = make_rsrc(do_base, (uint32_t)(rows_here - 1) * do_rowb + ROWB);
    const rsrc_t dq_rs = make_rsrc(dq_base, (uint32_t)(rows_here - 1) * dq_rowb + ROWB);
    const uint32_t o_rowb = (uint32_t)(p.o.row * 2);
    const rsrc_t o_rs = make_rsrc(o_base, (uint32_t)(rows_here - 1) * o_rowb + ROWB);
    const srd_t k_srd = make_srd(k_base, sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const srd_t v_srd = make_srd(v_base, sk > 0 ? (uint32_t)(sk - 1) * v_rowb + ROWB : 0u);

    int n_tiles = (sk + kDqBlockN - 1) / kDqBlockN;
    if (CAUSAL) {
        const int max_key = m0 + rows_here - 1 + delta;
        n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kDqBlockN + 1);
    }

    const int q_row = wave * 32 + l31;
    const int wave_q_lo = m0 + wave * 32, wave_q_hi = wave_q_lo + 31;

    // LDS-DMA staging (hand-issued, see fa_device.hpp:dma16_to_lds_hidden): wave w moves the DPW
    // 1-KiB pieces [w*DPW, (w+1)*DPW) of every K and V tile; swizzle applied to the source offset.
    constexpr int DPW = SLOTS / 8;
    uint32_t dma_goff_k[DPW], dma_goff_v[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int chunk = (wave * DPW + i) * 64 + lane, row = chunk / SLOTS, phys = chunk % SLOTS;
        const int slot = lds_tile_logical_slot<D>(row, phys);
        dma_goff_k[i] = row * k_rowb + slot * 16;
        dma_goff_v[i] = row * v_rowb + slot * 16;
    }
    const uint32_t lds_k0 = lds_addr(smem) + (uint32_t)wave * DPW * 1024;
    const uint32_t lds_v0 = lds_k0 + 2 * TILEB;
    auto dma_tiles = [&](int t, int buf) {                 // K(t), V(t) -> LDS buffers `buf`
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(k_srd, (uint32_t)(t * kDqBlockN) * k_rowb + dma_goff_k[i], lds_k0 + buf * TILEB + i * 1024);
            dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(v_srd, (uint32_t)(t * kDqBlockN) * v_rowb + dma_goff_v[i], lds_v0 + buf * TILEB + i * 1024);
        }
    };
    uint32_t row_rd[KS];       // row reads of K (for S^T) and V (for dP^T): same (row, slot) pattern
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) row_rd[ks] = lds_tile_off<D>(l31, 2 * ks + hi);
    uint32_t tr_rd[2][DB];     // transposed reads of K (A operand of dQ^T = K^T dS^T)
    {
        const int L = lane & 15, g = (lane >> 4) & 1;
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                tr_rd[sec][db] = lds_tile_off<D>(4 * hi + 8 * sec + (L >> 2), 4 * db + 2 * g + ((L & 3) >> 1)) + 8 * (L & 1);
    }

    // B operands held for the whole loop: Q^T and dO^T fragments of this lane's query row
    u32x4 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = buf_load16(q_rs, (uint32_t)q_row * q_rowb + (2 * ks + hi) * 16);
        dof[ks] = buf_load16(do_rs, (uint32_t)q_row * do_rowb + (2 * ks + hi) * 16);
    }
    // D_i = rowsum(dO_i * O_i) (flash_bwd_preprocess_kernel.h:23-96) is computed HERE, from the dO fragments this lane holds anyway and the
    // matching O fragments (one extra 64 KiB read per workgroup, hidden behind the first K / V tiles), and written to dsoftmax_sum for
    // the dK/dV launch that follows on the stream: the separate fa_bwd_dot_do_o launch (0.09 ms at C4, ~6 % of a 512-long backward) is
    // no longer on the path of fa_run_mha_bwd.  fp32 products and sums, like the stand-alone kernel.
    float lse2 = 0.f, dsum = 0.f;   // rows past the end keep 0 (they contribute nothing: Q = dO = 0)
    {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 of = buf_load16(o_rs, (uint32_t)q_row * o_rowb + (2 * ks + hi) * 16);      // rows past the end read zeros
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                part += LP<T>::to_float((uint16_t)(of[w] & 0xffff)) * LP<T>::to_float((uint16_t)(dof[ks][w] & 0xffff));
                part += LP<T>::to_float((uint16_t)(of[w] >> 16)) * LP<T>::to_float((uint16_t)(dof[ks][w] >> 16));
            }
        }
        dsum = sum_both_halves(part);           // a lane holds the even (hi = 0) or odd (hi = 1) 16-byte slots of its row
    }
    if (q_row < rows_here) {
        lse2 = p.lse_ptr[stat_off + q_row] * kLog2e;
        if (hi == 0) p.dsum_ptr[stat_off + q_row] = dsum;
    }
    const float c = p.scale_log2e;
    // D = 128: a loop-invariant block of -D_i feeds the first dP MFMA as its C operand, so dP - D costs no VALU (16 registers; the
    // D = 64 instance runs on 128 registers and has no room for it)
    constexpr bool kDpFromMinusD = D == 128;
    f32x16 negd;
#pragma unroll
    for (int r = 0; r < 16; ++r) negd[r] = -dsum;

    f32x16 dqacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[db][r] = 0.f;

    if (n_tiles > 0) dma_tiles(0, 0);
    // the Q / dO / LSE / D loads above are compiler-visible: force them home so the hand-counted
    // vmcnt(0) below also covers the DMA pieces
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));
    asm volatile("" : "+v"(lse2), "+v"(dsum));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // One key tile.  At D = 128 the loop below runs two tiles per trip, the ring slot is then a compile-time constant and every
    // LDS fragment read is base register + immediate (18 VALU address adds fewer per tile).
    auto tile_body = [&](int t, const int BUF) __attribute__((always_inline)) {
        const int n0 = t * kDqBlockN;
        FA_LDS char* kbuf = smem + BUF * TILEB;
        FA_LDS char* vbuf = smem + 2 * TILEB + BUF * TILEB;
        __syncthreads();      // tile t is in LDS (every wave waited for its pieces); the other buffer is free again
        // An LDS-DMA piece costs the issuing wave ~110 cycles (profiles/r3_dkdv_phase_timing.log): with FA_DQ_STAGGER_DMA waves 4-7 issue
        // theirs after the first 32-key half below, while waves 0-3 (which issue here) are in their MFMAs, instead of all eight at once
        const bool wave_active = !CAUSAL || (n0 <= wave_q_hi + delta);
        const bool dma_late = FA_DQ_STAGGER_DMA && wave >= 4 && wave_active;       // (a wave that skips this tile issues at once)
        if (t + 1 < n_tiles && !dma_late) dma_tiles(t + 1, BUF ^ 1);
        if (wave_active) {
            // Masking is 2 VALU per score element (compare with an immediate + select), not 4: the key index of
            // element (bi, r) is n0 + 32*bi + (r&3) + 8*(r>>2) + 4*hi, so everything lane- or tile-dependent
            // (n0, hi, the causal / sk limit) is folded ONCE per tile into `lim_loc`; per element only the compile-time
            // constant 32*bi + (r&3) + 8*(r>>2) is compared.  The selects sit behind a wave-uniform branch, so interior
            // tiles pay nothing: 12-17 % of the kernel (profiles/r2_bwd_valu_trims_ab.log).  (Two whole tile bodies behind
            // a branch spilled; a branch around the 32 selects alone does not.)
            const bool need_mask = (n0 + kDqBlockN > sk) || (CAUSAL && (n0 + kDqBlockN - 1 > wave_q_lo + delta));
            const int lim = CAUSAL ? min(sk - 1, m0 + q_row + delta) : sk - 1;
            const int lim_loc = lim - n0 - 4 * hi;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {           // two 32-key halves, keeps S/dP at 16+16 regs
                f32x16 sacc, dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = kDpFromMinusD ? negd[r] : 0.f; }   // dP chain from -D: dP - D comes out of the MFMAs
                // S^T and dP^T chains interleaved: consecutive MFMAs never share an accumulator; per-chain order unchanged.  (hipcc waits
                // for every pair of fragment reads right behind its issue here; requesting them 2-6 steps ahead by hand changed
                // nothing - the partner wave on the SIMD covers it - profiles/r2_bwd_dkdv_lds_ab.log.)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 kf = lds_read16(kbuf, row_rd[ks] + bi * 32 * ROWB);
                    const u32x4 vf = lds_read16(vbuf, row_rd[ks] + bi * 32 * ROWB);
                    sacc = LP<T>::mfma(kf, qf[ks], sacc);          // S^T = K Q^T
                    dpacc = LP<T>::mfma(vf, dof[ks], dpacc);       // dP^T = V dO^T
                }
                // P = exp(s*scale - LSE) (flash_bwd_kernel.h:474), dS = P * (dP - D) (:490)
                if constexpr (D == 128) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[r] = fast_exp2(__builtin_fmaf(sacc[r], c, -lse2));
                    if (need_mask) {                                // wave-uniform branch: interior tiles skip the 2 VALU per element
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[r] = (32 * bi + (r & 3) + 8 * (r >> 2)) <= lim_loc ? sacc[r] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[r] = sacc[r] * dpacc[r];
                } else {
                    // D = 64 (128 registers, two workgroups per CU): the branch and the -D block both cost spills there (measured
                    // +15-19 %); it keeps the select on every tile, the limit opened wide when the tile needs no mask
                    const int ll = need_mask ? lim_loc : 0x7fffffff;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float pv = fast_exp2(__builtin_fmaf(sacc[r], c, -lse2));
                        pv = (32 * bi + (r & 3) + 8 * (r >> 2)) <= ll ? pv : 0.f;
                        sacc[r] = pv * (dpacc[r] - dsum);
                    }
                }
                if (bi == 0 && dma_late && t + 1 < n_tiles) dma_tiles(t + 1, BUF ^ 1);
                // dQ^T (D x 32 queries) += K^T (D x 32 keys) * dS^T (32 keys x 32 queries)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const u32x4 dsf = pack_c_half<T>(sacc, half);   // dS rounded like the reference (:512)
                    const int ts = 2 * bi + half;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const u32x2 a0 = lds_read_tr8(kbuf, tr_rd[0][db] + ts * 16 * ROWB);
                        const u32x2 a1 = lds_read_tr8(kbuf, tr_rd[1][db] + ts * 16 * ROWB);
                        const u32x4 ktf = {a0.x, a0.y, a1.x, a1.y};
                        dqacc[db] = LP<T>::mfma(ktf, dsf, dqacc[db]);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t+1 have landed
    };
    {
        int t = 0;
        if constexpr (D == 128 && !FA_DQ_NO_UNROLL) {
            for (; t + 2 <= n_tiles; t += 2) {
                tile_body(t, 0);
                tile_body(t + 1, 1);
            }
            if (t < n_tiles) tile_body(t, 0);
        } else {                                  // D = 64 runs on 128 registers: one copy of the body, run-time slot
            for (; t < n_tiles; ++t) tile_body(t, t & 1);
        }
    }

    // epilogue: dQ *= scale (flash_bwd_kernel.h:765), round, stage through LDS, whole-row stores
    __syncthreads();
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            u32x2 w;
            w.x = LP<T>::pack2(dqacc[db][4 * g4 + 0] * p.scale, dqacc[db][4 * g4 + 1] * p.scale);
            w.y = LP<T>::pack2(dqacc[db][4 * g4 + 2] * p.scale, dqacc[db][4 * g4 + 3] * p.scale);
            lds_write8(smem, lds_tile_off<D>(q_row, 4 * db + g4) + 8 * hi, w);
        }
    __syncthreads();
    constexpr int O_CHUNKS = (kDqBlockM * SLOTS) / kDqThreads;
#pragma unroll
    for (int i = 0; i < O_CHUNKS; ++i) {
        const int chunk = tid + i * kDqThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
        buf_store16(dq_rs, (uint32_t)row * dq_rowb + slot * 16, lds_read16(smem, lds_tile_off<D>(row, slot)));
    }
}

// =============================================================================================
// dK/dV kernel: workgroup = 8 waves = 128 keys of one (batch, kv head); loops over the GQA
// group's query heads and over 64-row Q/dO tiles.  Wave w owns key block kb = w & 3 (32 keys)
// and the 32-row half qh = w >> 2 of every Q/dO tile; the two q-halves' partial dK^T / dV^T are
// summed through LDS once, in the epilogue.
//
// Register plan (lessons of the first two versions, see profiles/NOTEBOOK.md 3): the 128 long-lived
// accumulator registers live in AGPRs and are only touched by MFMAs (LP<T>::mfma_agpr, inline
// asm "+a"); with -amdgpu-mfma-vgpr-form every other MFMA result stays in VGPRs where the softmax
// VALU uses it directly, so there is no v_accvgpr shuttling; everything else fits 128 VGPRs, so
// TWO waves share each SIMD (a lone wave per SIMD left every LDS latency exposed: 39 % MFMA
// utilisation).  K^T / V^T operands are read from the workgroup's resident K/V tiles in LDS, Q/dO
// tiles arrive by LDS-DMA.
// LDS: K[128 keys] | V[128 keys] | Q ring[2] | dO ring[2] | stats ring[2]  = 129 KiB.
// =============================================================================================
// (workgroup shape kKvThreads / kKvBlockN / kKvBlockM and the epilogue: fa_bwd_dkdv_common.hpp.  D = 128 launches go to the two-group
// ping-pong kernel in fa_bwd_dkdv_pp.hip - same arithmetic, bit-identical results; this lock-step kernel serves D = 64, where two
// workgroups per CU already interleave four waves per SIMD, and stays instantiated at D = 128 as the A/B baseline: FA_DKDV_LOCKSTEP=1.)

// -DFA_KV_TIMING (development aid, tools/phase_timing_dkdv.py): per-wave s_memtime stamps at the phase edges of the tile loop, summed per
// phase and left in p.ws[(block * 8 + wave) * 8 + phase] (the caller passes a workspace; MHA launches never use it otherwise)
#ifdef FA_KV_TIMING
#define FA_KV_STAMP(i) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define FA_KV_STAMP(i) do { } while (0)
#endif
template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(kKvThreads, FA_KV_MIN_WAVES(D)) void fa_bwd_dkdv_kernel(const BwdKernelParams p) {
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int KVB = kKvBlockN * ROWB;                   // the workgroup's K (or V) tile
    constexpr int TILEB = kKvBlockM * ROWB;                 // one Q (or dO) tile
    constexpr int STATB = 2 * kKvBlockM * 4;                // lse2 + dsum of one tile
    constexpr int OFF_V = KVB, OFF_Q = 2 * KVB, OFF_DO = 2 * KVB + 2 * TILEB, OFF_STAT = 2 * KVB + 4 * TILEB;
    __shared__ __attribute__((aligned(16))) char smem_raw[OFF_STAT + 2 * STATB];   // the only LDS object
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* ktile = smem;
    FA_LDS char* vtile = smem + OFF_V;
    FA_LDS char* stat = smem + OFF_STAT;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave & 3, qh = wave >> 2;             // key block / q-half of this wave

    int tile, batch, vhead, tiles_seq;      // tile = 128-key block of this sequence (compact varlen grid: looked up in cu_seqlens_k)
    if (!decode_work<kKvBlockN>(blockIdx.x, p.n_k_tiles, p.varlen_slots, p.cu_seqlens_k, p.b, p.h_k * p.n_split, tile, batch, vhead, tiles_seq, p.group_heads)) return;
    // Few KV heads (GQA / MQA): the grid b * h_k * ceil(sk / 128) is small and, under a causal mask, unbalanced (the first key block
    // of a sequence sees every query tile, the last one a single tile).  With workspace from the caller the group's h / h_k query
    // heads are dealt to n_split workgroups; each leaves fp32 partial sums and fa_bwd_sum_splits_kernel adds them in a fixed order.
    const int head_k = vhead / p.n_split, split = vhead - head_k * p.n_split;
    const int heads_here = p.h_ratio / p.n_split;            // query heads of this workgroup

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch, v_boff = (int64_t)batch * p.v.batch,
            do_boff = (int64_t)batch * p.dout.batch, dk_boff = (int64_t)batch * p.dk.batch, dv_boff = (int64_t)batch * p.dv.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int qb = p.cu_seqlens_q[batch], kb = p.cu_seqlens_k[batch];
        sq = min(p.cu_seqlens_q[batch + 1] - qb, p.seqlen_q);   // clamp to the declared max_seqlen_q (padded LSE / D rows)
        sk = p.cu_seqlens_k[batch + 1] - kb;
        q_row0 = qb; k_row0 = kb;
        q_boff = k_boff = v_boff = do_boff = dk_boff = dv_boff = 0;
    }
    const int n0 = tile * kKvBlockN;
    if (n0 >= sk) return;
    const int delta = sk - sq;
    const int keys_here = min(kKvBlockN, sk - n0);

    const T* k_base = uniform_ptr((const T*)p.k_ptr + k_boff + (k_row0 + n0) * p.k.row + (int64_t)head_k * p.k.head);
    const T* v_base = uniform_ptr((const T*)p.v_ptr + v_boff + (k_row0 + n0) * p.v.row + (int64_t)head_k * p.v.head);
    T* dk_base = uniform_ptr((T*)p.dk_ptr + dk_boff + (k_row0 + n0) * p.dk.row + (int64_t)head_k * p.dk.head);
    T* dv_base = uniform_ptr((T*)p.dv_ptr + dv_boff + (k_row0 + n0) * p.dv.row + (int64_t)head_k * p.dv.head);
    const uint32_t k_rowb = (uint32_t)(p.k.row * 2), v_rowb = (uint32_t)(p.v.row * 2), q_rowb = (uint32_t)(p.q.row * 2), do_rowb = (uint32_t)(p.dout.row * 2);
    const srd_t k_srd = make_srd(k_base, (uint32_t)(keys_here - 1) * k_rowb + ROWB);
    const srd_t v_srd = make_srd(v_base, (uint32_t)(keys_here - 1) * v_rowb + ROWB);

    // Q-tile range: key j is visible to query i iff i >= j - delta
    const int n_q_tiles = (sq + kKvBlockM - 1) / kKvBlockM;
    int qt_begin = 0;
    if (CAUSAL) qt_begin = max(0, n0 - delta) / kKvBlockM;
    const int tiles_per_head = max(0, n_q_tiles - qt_begin);
    const int n_iters = tiles_per_head * heads_here;

    const int key_row = kb * 32 + l31;                   // this lane's key inside the 128-key block
    const int wave_k_lo = n0 + kb * 32, wave_k_hi = wave_k_lo + 31;

    // ---- LDS-DMA tables: a tile is 1 KiB pieces (64 lanes x 16 B, lane-linear in LDS, swizzle on
    // the source offset); wave w moves pieces [w*PPW, (w+1)*PPW) ------------------------------------
    constexpr int PPW_KV = (kKvBlockN * SLOTS / 64) / 8;  // pieces per wave: K/V tile (4 for d=128)
    constexpr int PPW_Q = (kKvBlockM * SLOTS / 64) / 8;   //                  Q/dO tile (2 for d=128)
    auto piece_src = [&](int piece, uint32_t rowb) {       // global byte offset of this lane's chunk of `piece`
        const int chunk = piece * 64 + lane, row = chunk / SLOTS, phys = chunk % SLOTS;
        return (uint32_t)row * rowb + lds_tile_logical_slot<D>(row, phys) * 16;
    };
    const uint32_t lds0 = lds_addr(smem);
    uint32_t q_src[PPW_Q], do_src[PPW_Q];                  // per-lane source offsets of this wave's Q/dO pieces
#pragma unroll
    for (int i = 0; i < PPW_Q; ++i) {
        q_src[i] = piece_src(wave * PPW_Q + i, q_rowb);
        do_src[i] = piece_src(wave * PPW_Q + i, do_rowb);
    }

    // rows read as MFMA operands with 8 contiguous d per lane (row reads): row l31 (+32*block), slot 2*ks+hi
    uint32_t row_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) row_rd[ks] = lds_tile_off<D>(l31, 2 * ks + hi);
    uint32_t tr_rd[2][DB];
    {
        const int L = lane & 15, g = (lane >> 4) & 1;
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                tr_rd[sec][db] = lds_tile_off<D>(4 * hi + 8 * sec + (L >> 2), 4 * db + 2 * g + ((L & 3) >> 1)) + 8 * (L & 1);
    }
    const float c = p.scale_log2e;

    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[db][r] = 0.f; dvacc[db][r] = 0.f; }

    // The Q / dO / statistics streams of ONE query head are addressed through descriptors that are rebuilt only when the stream moves on
    // to the next head of the group; inside a head a tile costs one scalar multiply and one VALU add per DMA piece.  (Until round 3
    // the descriptors were rebuilt for every tile: ~110 SALU instructions at the top of every iteration, 600-1000 cycles per tile with
    // the matrix pipe idle - both waves of a SIMD do it at the same time; tools/phase_timing_dkdv.py, profiles/r3_dkdv_phase_timing.log.)
    const int head_first = head_k * p.h_ratio + split * heads_here;
    srd_t q_srd = make_srd(nullptr, 0), do_srd = make_srd(nullptr, 0);
    rsrc_t st_rs = make_rsrc(nullptr, 0);
    auto set_head = [&](int hq) {
        const T* qb = uniform_ptr((const T*)p.q_ptr + q_boff + q_row0 * p.q.row + (int64_t)hq * p.q.head);
        const T* dob = uniform_ptr((const T*)p.do_ptr + do_boff + q_row0 * p.dout.row + (int64_t)hq * p.dout.head);
        q_srd = make_srd(qb, sq > 0 ? (uint32_t)(sq - 1) * q_rowb + ROWB : 0u);          // rows past the end of the sequence read zeros
        do_srd = make_srd(dob, sq > 0 ? (uint32_t)(sq - 1) * do_rowb + ROWB : 0u);
        // the 64 LSE / D values of a tile go through ONE register of waves 0 / 1 (other waves: zero-record descriptor, no access)
        const float* sb = uniform_ptr((wave == 0 ? p.lse_ptr : p.dsum_ptr) + ((int64_t)batch * p.h + hq) * p.lse_row_stride);
        st_rs = make_rsrc(sb, wave < 2 ? (uint32_t)sq * 4u : 0u);
    };
    // prefetch cursor: (query head, tile inside the head) of the next tile to request
    int pf_head = head_first, pf_tile = 0;
    auto pf_m0 = [&]() { return (qt_begin + pf_tile) * kKvBlockM; };
    auto pf_advance = [&]() {
        if (++pf_tile == tiles_per_head) { pf_tile = 0; ++pf_head; if (pf_head < head_first + heads_here) set_head(pf_head); }
    };
    // Q/dO tile at the cursor -> ring slot buf by LDS-DMA issued from inline asm: nothing returns to a VGPR, hipcc has no reason to wait.
    auto issue_tile = [&](int buf) {
        const uint32_t m0 = (uint32_t)pf_m0();
#pragma unroll
            for (int i = 0; i < PPW_Q; ++i) {
                const int piece = wave * PPW_Q + i;
                dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(q_srd, q_src[i] + m0 * q_rowb, lds0 + OFF_Q + buf * TILEB + piece * 1024);
                dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(do_srd, do_src[i] + m0 * do_rowb, lds0 + OFF_DO + buf * TILEB + piece * 1024);
            }
    };
    // Statistics of the tile at the cursor: loaded when the tile's DMA is issued, transformed (-LSE*log2e, -D) and written to the stats
    // slot at the END of the same iteration, in front of the vmcnt(0) that is there anyway.  The consumers then need no per-element
    // multiply / subtract: exp2(fma(s, c, nl)) and a dP chain that starts from -D (16 VALU fewer per wave-tile each).  The load is
    // unconditional and its register is consumed on every path.  History: in round 1 the statistics went through a register under an
    // exec-masked branch and hipcc answered with `s_waitcnt vmcnt(0)` at the top of every iteration - a write-after-write guard for
    // the paths that skipped the branch - right behind the DMA issue, exposing the L2 / HBM latency of the NEXT tile in front of the
    // MFMAs of the current one (58 % of wave cycles parked, 35 % MFMA busy).
    const float st_mult = wave == 0 ? -kLog2e : -1.0f;
    auto load_stat = [&](bool valid) -> float {     // every wave, every iteration; !valid: an offset past every descriptor's range (returns 0)
        const uint32_t off = valid ? ((uint32_t)pf_m0() + (uint32_t)lane) * 4u : 0xfffffff0u;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(st_rs, off, 0, 0));
    };
    auto store_stat = [&](float x, int buf) {
        *(FA_LDS float*)(stat + buf * STATB + wave * (kKvBlockM * 4) + lane * 4) = x * st_mult;
    };

    // ---- prologue: this workgroup's K and V tiles + the first Q/dO tile ---------------------------
#pragma unroll
    for (int i = 0; i < PPW_KV; ++i) {
        const int piece = wave * PPW_KV + i;
        dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(k_srd, piece_src(piece, k_rowb), lds0 + piece * 1024);
        dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(v_srd, piece_src(piece, v_rowb), lds0 + OFF_V + piece * 1024);
    }
    if (n_iters > 0) {
        set_head(pf_head);
        issue_tile(0);
        if (wave < 2) store_stat(load_stat(true), 0);
        pf_advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // This wave's K (and V) fragments of the first KREG (VREG) k-steps stay in registers for the whole loop: they never change, and the
    // kernel is LDS-bandwidth-bound - every S / dP MFMA used to read BOTH its operands from LDS, 384 KiB per workgroup-tile against
    // 2048 MFMA cycles per SIMD.  -5 % (C4) to -13 % (8k causal), bit-identical.
    constexpr int KREG = FA_KV_KREG(D) < KS ? FA_KV_KREG(D) : KS, VREG = FA_KV_VREG(D) < KS ? FA_KV_VREG(D) : KS;
    u32x4 kreg[KREG > 0 ? KREG : 1], vreg[VREG > 0 ? VREG : 1];
#pragma unroll
    for (int ks = 0; ks < KREG; ++ks) kreg[ks] = lds_read16(ktile, row_rd[ks] + kb * 32 * ROWB);
#pragma unroll
    for (int ks = 0; ks < VREG; ++ks) vreg[ks] = lds_read16(vtile, row_rd[ks] + kb * 32 * ROWB);

#ifdef FA_KV_TIMING
    uint64_t tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    // (a static s_setprio 1 for the younger q-half, which buys the 16x16x32 dK/dV kernel 1-3 %, measures +-0.2 % here and in both dQ kernels:
    // profiles/r4_bwd_prio_dq_dkdv32_ab.log)
    int cur_tile = 0;                                       // tile-in-head index of the tile being computed
    for (int it = 0; it < n_iters; ++it) {
        const int m0 = (qt_begin + cur_tile) * kKvBlockM;
        if (++cur_tile == tiles_per_head) cur_tile = 0;
        const int buf = it & 1;
        FA_LDS char* qbuf = smem + OFF_Q + buf * TILEB;
        FA_LDS char* dobuf = smem + OFF_DO + buf * TILEB;
        FA_LDS char* sbuf = stat + buf * STATB;
        const bool more = (it + 1 < n_iters);
        FA_KV_STAMP(5);                                     // (loop edge + barrier exit)
        if (more && (!FA_KV_STAGGER_DMA || qh == 0)) issue_tile(buf ^ 1);          // ring slot buf^1 was last read in iteration it-1
        const float st_next = load_stat(more);
        if (!FA_KV_STAGGER_DMA && more) pf_advance();
        FA_KV_STAMP(0);                                     // DMA issue

        // wave-level causal skip: all 32 rows of this wave's half are above the diagonal for all its 32 keys
        const int mh = m0 + 32 * qh;                       // first query row of this wave's half
        // No wave-level causal skip here, on purpose.  `if (wave_active) { body }` around a body whose accumulators are
        // inline-asm "+a" operands makes hipcc copy all 128 accumulator registers out and back every iteration (258
        // v_accvgpr moves + 48 scratch ops in the D=128 loop, 147 moves at D=64).  A fully masked wave-tile instead runs the
        // body with thr past every row: P = dS = 0 by select, the MFMAs add exact zeros (outputs bit-identical).  Measured,
        // whole backward, causal (tools/ab_bwd.py, profiles/r1_bwd_causal_ab.log): 0.60x at 8k/16k, 0.69x at 1k, D=64 0.89x.
#ifdef FA_TEST_DKDV_SKIP_BRANCH   // tests/test_kernel_resources_cpu.py only: reinstates the branch so the guard can prove it detects it
        if (!CAUSAL || (wave_k_lo <= mh + 31 + delta))
#endif
        {
            // causal mask, 2 VALU per element: element r = 4*g4 + e is query row m0 + 32*qh + 8*g4 + 4*hi + e, visible iff
            // key <= row + delta  <=>  8*g4 + e >= thr with everything tile- / lane-dependent folded into `thr` once per
            // tile (INT_MIN when this wave's tile needs no mask); the left side is a compile-time constant.
            const bool need_mask = CAUSAL && (wave_k_hi > mh + delta);
            const int thr = (n0 + key_row) - (mh + 4 * hi + delta);
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                        // the dP chain starts from -D (registers 4*g4.. = rows 32*qh + 8*g4 + 4*hi + {0..3})
                const f32x4 nd4 = *(const FA_LDS f32x4*)(sbuf + kKvBlockM * 4 + (32 * qh + 8 * g4 + 4 * hi) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dpacc[4 * g4 + e] = nd4[e];
            }
            // S and dP chains interleaved (consecutive MFMAs on different accumulators, see fa_bwd_dq_kernel).  hipcc leaves these reads
            // one MFMA of look-ahead (`ds_read x2; s_waitcnt lgkmcnt(2); v_mfma`); requesting them further ahead by hand, as the dQ
            // kernel does, costs address-register spills here at every depth tried (1-4 steps: 10-26 scratch loads per tile, each
            // followed by vmcnt(0)): the kernel has 128 VGPRs next to its 128 accumulator registers and they are all spoken for.
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 qa = lds_read16(qbuf, row_rd[ks] + qh * 32 * ROWB);
                const u32x4 kf = ks < KREG ? kreg[ks < KREG ? ks : 0] : lds_read16(ktile, row_rd[ks] + kb * 32 * ROWB);
                const u32x4 da = lds_read16(dobuf, row_rd[ks] + qh * 32 * ROWB);
                const u32x4 vf = ks < VREG ? vreg[ks < VREG ? ks : 0] : lds_read16(vtile, row_rd[ks] + kb * 32 * ROWB);
                sacc = LP<T>::mfma(qa, kf, sacc);                   // S = Q K^T  (rows = queries, lane = key)
                dpacc = LP<T>::mfma(da, vf, dpacc);                 // dP = dO V^T
            }
            FA_KV_STAMP(1);                                 // S / dP MFMAs
            if constexpr (FA_KV_STAGGER_DMA) {    // waves 4-7 request their pieces HERE, while waves 0-3 are still in their S / dP MFMAs
                if (more && qh == 1) issue_tile(buf ^ 1);
                if (more) pf_advance();
            }
            f32x16 pacc;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 nl4 = *(const FA_LDS f32x4*)(sbuf + (32 * qh + 8 * g4 + 4 * hi) * 4);      // -LSE * log2(e) of the same rows
#pragma unroll
                for (int e = 0; e < 4; ++e) pacc[4 * g4 + e] = fast_exp2(__builtin_fmaf(sacc[4 * g4 + e], c, nl4[e]));
            }
            if (need_mask) {                                        // wave-uniform branch: only diagonal tiles pay for the select
#pragma unroll
                for (int r = 0; r < 16; ++r) pacc[r] = (8 * (r >> 2) + (r & 3) >= thr) ? pacc[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = pacc[r] * dpacc[r];     // dS = P * (dP - D) (flash_bwd_kernel.h:1354)
            // dV^T += dO^T P and dK^T += Q^T dS: 4*DB MFMAs whose A operands are transposed LDS reads.  The accumulators are inline-asm
            // operands, so hipcc has no latency model for these MFMAs and used to issue each read pair right in front of its
            // consumer (s_waitcnt lgkmcnt(0) before every MFMA: the LDS latency was exposed 4*DB times per tile).  The reads are
            // software-pipelined by hand instead - fragment j + PF is requested before MFMA j - and the order is pinned.
            u32x4 pfr[2], dsfr[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                pfr[half] = pack_c_half<T>(pacc, half);             // P rounded (flash_bwd_kernel.h:1359)
                dsfr[half] = pack_c_half<T>(sacc, half);            // dS rounded (:1360)
            }
            FA_KV_STAMP(2);                                 // exp / mask / dS / pack
            constexpr int NST = 4 * DB, PF = FA_KV_PF2(D);                  // step j = (half, db, which): which 0 -> dV (dO^T), 1 -> dK (Q^T)
            auto rd_frag = [&](int j) {
                const int half = j / (2 * DB), db = (j >> 1) % DB, ts = 2 * qh + half;
                FA_LDS char* src = (j & 1) ? qbuf : dobuf;
                const u32x2 a0 = lds_read_tr8(src, tr_rd[0][db] + ts * 16 * ROWB);
                const u32x2 a1 = lds_read_tr8(src, tr_rd[1][db] + ts * 16 * ROWB);
                return u32x4{a0.x, a0.y, a1.x, a1.y};
            };
            u32x4 frag[NST];
#pragma unroll
            for (int j = 0; j < PF; ++j) frag[j] = rd_frag(j);
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                if (j + PF < NST) frag[j + PF] = rd_frag(j + PF);
                __builtin_amdgcn_sched_barrier(0);
                const int half = j / (2 * DB), db = (j >> 1) % DB;
                if (j & 1) LP<T>::mfma_agpr(dkacc[db], frag[j], dsfr[half]);      // dK^T += Q^T dS    (AGPR accumulator)
                else LP<T>::mfma_agpr(dvacc[db], frag[j], pfr[half]);             // dV^T += dO^T P    (AGPR accumulator)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        FA_KV_STAMP(3);                                     // dV / dK MFMAs
            if (more && wave < 2) store_stat(st_next, buf ^ 1);
            asm volatile("" :: "v"(st_next));               // consumed on every path: hipcc never has to guard the register at the loop top
            // (timing-only ablations, profiles/r2_bwd_dkdv_lds_ab.log: without this wait -0.2 %, without wait AND barrier -5..-6 %)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's DMA pieces (and statistics) have landed
            FA_KV_STAMP(4);                                 // vmcnt wait
        __syncthreads();
    }
#ifdef FA_KV_TIMING
    if (p.ws != nullptr && lane == 0) {
        float* out = p.ws + ((int64_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) out[i] = (float)tacc[i];
        out[6] = (float)n_iters;
    }
#endif

    // ---- epilogue (fa_bwd_dkdv_common.hpp): the loop's last barrier has passed, K/V tiles, rings and stats are dead, LDS is scratch ----
    dkdv_epilogue<T, D, OFF_STAT>(p, smem, dkacc, dvacc, batch, head_k, split, k_row0, n0, keys_here, dk_base, dv_base);
}

// Sum of the n_split fp32 partial dK / dV planes (fixed order: deterministic), softmax scale on dK (flash_bwd_kernel.h:1652-1654),
// rounding, strided store.  One thread = 8 consecutive d of one (key row, kv head); HBM-bound, 2 * n_split * 32 + 2 * 16 bytes each.
constexpr int kSumThreads = 256;
template <typename T, int D>
__global__ __launch_bounds__(kSumThreads) void fa_bwd_sum_splits_kernel(const BwdKernelParams p) {
    constexpr int CH = D / 8;                                                   // 16-byte output chunks per (row, head)
    const int64_t n_items = p.ws_rows * p.h_k * CH;
    const int64_t item = (int64_t)blockIdx.x * kSumThreads + threadIdx.x;
    if (item >= n_items) return;
    const int ch = (int)(item % CH);
    const int64_t rh = item / CH;
    const int hk = (int)(rh % p.h_k);
    const int64_t row = rh / p.h_k;                                            // key row of the whole batch
    const int64_t plane = p.ws_rows * p.h_k * D;
    int64_t out_row = row, out_batch = 0;
    if (p.cu_seqlens_k == nullptr) { out_batch = row / p.seqlen_k; out_row = row - out_batch * p.seqlen_k; }
    else if (row >= p.cu_seqlens_k[p.b]) return;                               // padding rows of a packed tensor: no workgroup wrote them
#pragma unroll
    for (int tensor = 0; tensor < 2; ++tensor) {
        const float* src = p.ws + (int64_t)tensor * p.n_split * plane + rh * D + ch * 8;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.n_split; ++s) {
            a += *(const f32x4*)(src + (int64_t)s * plane);
            b += *(const f32x4*)(src + (int64_t)s * plane + 4);
        }
        const float mult = tensor == 0 ? p.scale : 1.0f;
        u32x4 w;
        w.x = LP<T>::pack2(a[0] * mult, a[1] * mult); w.y = LP<T>::pack2(a[2] * mult, a[3] * mult);
        w.z = LP<T>::pack2(b[0] * mult, b[1] * mult); w.w = LP<T>::pack2(b[2] * mult, b[3] * mult);
        const TStride& st = tensor == 0 ? p.dk : p.dv;
        T* dst = (T*)(tensor == 0 ? p.dk_ptr : p.dv_ptr) + out_batch * st.batch + out_row * st.row + (int64_t)hk * st.head + ch * 8;
        *(u32x4*)dst = w;
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int D>
static hipError_t launch_dot_t(const BwdKernelParams& kp, hipStream_t s) {
    const uint32_t tiles = (uint32_t)((kp.seqlen_q + kDotRowsPerBlock - 1) / kDotRowsPerBlock);
    dim3 grid(kp.varlen_slots != 0 ? kp.varlen_slots : tiles, kp.h, kp.varlen_slots != 0 ? 1 : kp.b);
    hipLaunchKernelGGL((fa_bwd_dot_do_o_kernel<T, D>), grid, dim3(kDotThreads), 0, s, kp);
    return hipGetLastError();
}
template <typename T, int D>
static hipError_t launch_dq_t(const BwdKernelParams& kp, hipStream_t s) {
    const uint32_t grid = kp.varlen_slots != 0 ? kp.varlen_slots * (uint32_t)kp.h : kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    if (grid == 0) return hipSuccess;
    if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, true>), dim3(grid), dim3(kDqThreads), 0, s, kp);
    else hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, false>), dim3(grid), dim3(kDqThreads), 0, s, kp);
    return hipGetLastError();
}
hipError_t launch_bwd_dkdv16(const BwdKernelParams& kp, int dtype, uint32_t grid, hipStream_t s);      // fa_bwd_dkdv16.hip (head_dim 128, v_mfma_f32_16x16x32)
static bool bwd_use_mfma16(const BwdKernelParams& kp, bool dkdv);                                      // (below, next to the launchers)
template <typename T, int D>
static hipError_t launch_dkdv_t(const BwdKernelParams& kp, hipStream_t s) {
    const uint32_t grid = (kp.varlen_slots != 0 ? kp.varlen_slots * (uint32_t)kp.h_k : kp.n_k_tiles * (uint32_t)kp.b * (uint32_t)kp.h_k) * (uint32_t)kp.n_split;
    if (grid == 0) return hipSuccess;
    hipError_t e;
    if (bwd_use_mfma16(kp, true)) {
        e = launch_bwd_dkdv16(kp, std::is_same<T, _Float16>::value ? 0 : 1, grid, s);
    } else {
        if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        else hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        e = hipGetLastError();
    }
    if (e != hipSuccess || kp.n_split == 1) return e;
    const int64_t items = kp.ws_rows * kp.h_k * (D / 8);
    hipLaunchKernelGGL((fa_bwd_sum_splits_kernel<T, D>), dim3((uint32_t)((items + kSumThreads - 1) / kSumThreads)), dim3(kSumThreads), 0, s, kp);
    return hipGetLastError();
}

#define FA_DISPATCH(FN, kp, dtype, s)                                                              \
    ((dtype) == 0 ? ((kp).d == 128 ? FN<_Float16, 128>(kp, s) : FN<_Float16, 64>(kp, s))           \
                  : ((kp).d == 128 ? FN<__bf16, 128>(kp, s) : FN<__bf16, 64>(kp, s)))

hipError_t launch_bwd_dot_do_o(BwdKernelParams kp, int dtype, hipStream_t s) {
    const uint32_t tiles = (uint32_t)((kp.seqlen_q + kDotRowsPerBlock - 1) / kDotRowsPerBlock);
    kp.varlen_slots = kp.cu_seqlens_q != nullptr ? varlen_slot_count(kp.total_q, kp.b, kDotRowsPerBlock, tiles) : 0u;
    return FA_DISPATCH(launch_dot_t, kp, dtype, s);
}
// head_dim 128 has a second set of kernels, re-tiled for v_mfma_f32_16x16x32 (fa_bwd_dq16.hip, fa_bwd_dkdv16.hip; why: fa_fwd_pp16.hip).
// Measured against the kernels of this file (profiles/r3_bwd_mfma16_ab.log, three boxes): dQ -1..-6 % without a mask at every length but
// +1..+4 % under a causal mask; dK/dV -2..-8 % from 1k x 1k on, +-1.5 % at 512.  FA_POLICY_AUTO follows those signs; fa_set_kernel_policy()
// pins either set.
constexpr int64_t kKvMfma16MinPairs = (int64_t)1 << 20;       // seqlen_q * seqlen_k per head: 1024 x 1024, causal or not, since its ring addressing
                                                              // costs one XOR per base register and tile (profiles/r3_policy_sweep.log, second table:
                                                              // -2..-8 % from 1k on, +-1.5 % at 512); per head, not per launch: fa_fwd_pp.hip says why
// head_dim 64 (round 5): both backward kernels also exist in the 16x16x32 tiling (fa_bwd_dq16.hip with 128-key tiles, fa_bwd_dkdv16.hip), one workgroup
// per compute unit on 160-180 registers against two co-resident workgroups of the 32x32x16 kernels on 128.  Measured, interleaved, two boxes
// (profiles/r5_bwd_d64_mfma16_ab.log, time 16 / 32 at b4 h32): dQ without a mask 0.92 / 0.96 / 0.92 / 0.91 / 0.89 / 0.89 at 512 .. 16k, under a causal mask
// 0.96 / 1.03 / 0.99 / 0.92-0.96 / 0.90-0.92 at 1k .. 16k (b16 x 2k: 1.12-1.18, sq 8k sk 1k: 1.17: two co-resident workgroups hide each other's prologue and
// epilogue and balance uneven tiles); dK/dV without a mask 0.97 / 1.01 / 0.97 / 0.95 / 0.94 / 0.95, under one 1.02 / 1.06 / 1.04 / 1.00-1.04 / 0.96-1.00.
// FA_POLICY_AUTO follows those signs per head: seqlen_q * seqlen_k from which the 16x16x32 kernel serves (0 = never).
#ifndef FA_BWD_D64_DQ16_MIN_PAIRS
#define FA_BWD_D64_DQ16_MIN_PAIRS ((int64_t)1 << 18)      // 512 x 512: the shortest sequence of the A/B above (until round 6 every non-causal dQ down to 1 x 1 went to the one-workgroup-per-unit kernel unmeasured)
#endif
#ifndef FA_BWD_D64_DQ16_MIN_PAIRS_CAUSAL
#define FA_BWD_D64_DQ16_MIN_PAIRS_CAUSAL ((int64_t)1 << 26)
#endif
#ifndef FA_BWD_D64_DKDV16_MIN_PAIRS
#define FA_BWD_D64_DKDV16_MIN_PAIRS ((int64_t)1 << 24)
#endif
#ifndef FA_BWD_D64_DKDV16_MIN_PAIRS_CAUSAL
#define FA_BWD_D64_DKDV16_MIN_PAIRS_CAUSAL ((int64_t)1 << 28)
#endif
static bool bwd_use_mfma16(const BwdKernelParams& kp, bool dkdv) {
    const int policy = kernel_policy();
    if ((kp.d != 128 && kp.d != 64) || policy == 0) return false;
    if (policy == 1) return true;
    if (kp.d == 64) {
        // (seqlen_k < seqlen_q under a mask: blocks of dead rows, where the narrow kernels are ahead)
        if (kp.is_causal && kp.seqlen_k < kp.seqlen_q) return false;
        // Round 6: the per-head thresholds of round 5 hold for launches that fill the chip; a launch that leaves the second workgroup slot of the 32x32x16 kernels empty is
        // 5-25 % faster on the one-workgroup 16x16x32 kernels at every length (ratio 16 / 32, fp16, b1 h8 / b1 h32: dQ 0.75-0.90, dK/dV 0.81-0.96), and a launch of many short
        // sequences (b16 h32 s512-1k) is 3-24 % faster on the 32x32x16 dQ: profiles/r6_policy_d64_before.log / r6_policy_d64.log.
        const int64_t pairs = (int64_t)kp.seqlen_q * kp.seqlen_k, cus = device_cu_count();
        if (dkdv) {
            const int64_t wgs = policy_bh(kp.b, kp.h) * (((int64_t)kp.seqlen_k + 127) / 128);      // (query heads: the unit of work, and what a shard states)
            if (kp.is_causal) return pairs >= (int64_t)FA_BWD_D64_DKDV16_MIN_PAIRS_CAUSAL || wgs <= 2 * cus || (wgs <= 4 * cus && pairs >= ((int64_t)1 << 24));
            return pairs >= (int64_t)FA_BWD_D64_DKDV16_MIN_PAIRS || wgs <= 2 * cus;
        }
        const int64_t wgs = policy_bh(kp.b, kp.h) * (((int64_t)kp.seqlen_q + 255) / 256);
        if (kp.is_causal) return pairs >= (int64_t)FA_BWD_D64_DQ16_MIN_PAIRS_CAUSAL || wgs <= 2 * cus;
        return pairs >= (int64_t)FA_BWD_D64_DQ16_MIN_PAIRS && !(wgs >= 4 * cus && pairs < ((int64_t)1 << 22));
    }
    // dQ (round 6): like the forward, by whether the launch fills the chip - from one 256-row workgroup per compute unit without a mask, from two under one (and then
    // from 1k x 1k).  Ratio 16 / 32 (profiles/r6_policy_small_grids.log, r5_policy_sweep_after_trim.log): no mask b1 h8 1.05 at 512-4k (16-128 workgroups), 0.92 at 8k
    // (256); b1 h32 1.03-1.04 at 512-1k (64-128), 0.935 at 2k (256); b4 h32 0.96 at 512 (256).  Causal: b1 h8 1.06-1.08 up to 4k (128), 1.03 at 8k (256); b1 h32 1.02 at
    // 2k (256), 0.96 at 4k (512), 0.95 at 8k; b4 h32 1.01 at 512 (256), 0.98 at 1k (512), 0.97-0.98 up to 8k.  (Rounds 3-5: always without a mask, from 2^28 pairs under one.)
    if (!dkdv) {
        const int64_t wgs = policy_bh(kp.b, kp.h) * (((int64_t)kp.seqlen_q + 255) / 256), cus = device_cu_count();
        return kp.is_causal ? (wgs >= 2 * cus && (int64_t)kp.seqlen_q * kp.seqlen_k >= kKvMfma16MinPairs) : wgs >= cus;
    }
    // dK/dV: from 1k x 1k whatever the launch (0.91-0.98 on every grid measured), and from 512 x 512 when the launch brings two 128-key workgroups per compute unit
    // (b4 h32 s512: 0.966; b1 h8: 1.00-1.01, b1 h32 / b2 h16: 0.98: profiles/r6_policy_small_grids.log)
    const int64_t pairs = (int64_t)kp.seqlen_q * kp.seqlen_k;
    return pairs >= kKvMfma16MinPairs || (pairs >= ((int64_t)1 << 18) && policy_bh(kp.b, kp.h) * (((int64_t)kp.seqlen_k + 127) / 128) >= 2 * device_cu_count());
}
const char* bwd_kernel_name_for(const BwdKernelParams& kp, bool dkdv) {
    return dkdv ? (bwd_use_mfma16(kp, true) ? "fa_bwd_dkdv16_kernel" : "fa_bwd_dkdv_kernel") : (bwd_use_mfma16(kp, false) ? "fa_bwd_dq16_kernel" : "fa_bwd_dq_kernel");
}
hipError_t launch_bwd_dq16(const BwdKernelParams& kp, int dtype, hipStream_t s);      // fa_bwd_dq16.hip (head_dim 128, v_mfma_f32_16x16x32)
hipError_t launch_bwd_dq(BwdKernelParams kp, int dtype, hipStream_t s) {
    kp.n_q_tiles = (uint32_t)((kp.seqlen_q + kDqBlockM - 1) / kDqBlockM);
    kp.varlen_slots = kp.cu_seqlens_q != nullptr ? varlen_slot_count(kp.total_q, kp.b, kDqBlockM, kp.n_q_tiles) : 0u;
    kp.group_heads = causal_group_heads(kp.is_causal != 0, kp.varlen_slots != 0 ? kp.b : 0, kp.varlen_slots != 0 ? kp.h : (int64_t)kp.b * kp.h, kp.seqlen_q, kp.seqlen_k, kp.n_q_tiles, (kp.d == 64 && !bwd_use_mfma16(kp, false)) ? 2 : 1, (int64_t)4 * kp.seqlen_k * kp.d);
    if (bwd_use_mfma16(kp, false)) return launch_bwd_dq16(kp, dtype, s);
    return FA_DISPATCH(launch_dq_t, kp, dtype, s);
}
#ifndef FA_KV_SPLIT_CAUSAL_PER_CU
#define FA_KV_SPLIT_CAUSAL_PER_CU 4
#endif
// Split rule: double the split while the group divides evenly and the launch has fewer workgroups than 4 per CU (causal: the key
// blocks of a sequence carry 1 .. n tiles of work, dynamic dispatch needs several workgroups per CU to even that out) or 1 per CU
// (no mask: equal work, only an underfilled chip gains); packed tensors need total_k to size the planes.
static int64_t dkdv_rows(const BwdKernelParams& kp) { return kp.cu_seqlens_k != nullptr ? kp.total_k : (int64_t)kp.b * kp.seqlen_k; }
int64_t dkdv_workspace_bytes(const BwdKernelParams& kp, int32_t n_split) {
    return n_split <= 1 ? 0 : 2 * (int64_t)n_split * dkdv_rows(kp) * kp.h_k * kp.d * 4;
}
// CUs of the CURRENT device (the split target is "workgroups per CU"), cached per device id (a process may drive devices or partitions with
// different CU counts: ADVICE r3); 256 = MI355X when there is no usable device (host-only callers such as fa_bwd_workspace_bytes on a build box).
// Consequence, stated in the public header: the summation order of GQA / MQA dK / dV with a workspace depends on the device's CU count.
int64_t device_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
        (void)hipGetLastError();                         // a box without a GPU: do not leave a sticky error behind
        return 256;
    }
    if (dev < 64) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return 256;
    }
    if (dev < 64) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}
int32_t dkdv_split(const BwdKernelParams& kp, int64_t avail_bytes) {
    if (kp.h_ratio <= 1 || dkdv_rows(kp) <= 0) return 1;
    const int64_t n_k_tiles = (kp.seqlen_k + kKvBlockN - 1) / kKvBlockN;
    // workgroups of the unsplit launch, exactly as launch_dkdv_t sizes its grid: varlen_slot_count() == 0 means "the plain
    // tiles x batch grid" (uniform lengths, batches past kVarlenMaxBatch), not "no workgroups"
    const int64_t slots = kp.cu_seqlens_k != nullptr ? (int64_t)varlen_slot_count(kp.total_k, kp.b, kKvBlockN, (uint32_t)n_k_tiles) : 0;
    const int64_t wgs = (slots != 0 ? slots : n_k_tiles * kp.b) * kp.h_k;
    const int64_t cus = device_cu_count();
    const int64_t want = kp.is_causal ? FA_KV_SPLIT_CAUSAL_PER_CU * cus : cus;
    int32_t split = 1;
    while (split * 2 <= kp.h_ratio && kp.h_ratio % (split * 2) == 0 && wgs * split < want &&
           (avail_bytes < 0 || dkdv_workspace_bytes(kp, split * 2) <= avail_bytes))
        split *= 2;
    return split;
}
hipError_t launch_bwd_dkdv(BwdKernelParams kp, int dtype, hipStream_t s) {
    kp.n_k_tiles = (uint32_t)((kp.seqlen_k + kKvBlockN - 1) / kKvBlockN);
    kp.varlen_slots = kp.cu_seqlens_k != nullptr ? varlen_slot_count(kp.total_k, kp.b, kKvBlockN, kp.n_k_tiles) : 0u;
    kp.n_split = kp.ws != nullptr ? dkdv_split(kp, kp.ws_bytes) : 1;
    kp.ws_rows = dkdv_rows(kp);
    // (key block 0 is the heaviest under a causal mask: ascending tile order is heaviest first already)
    kp.group_heads = causal_group_heads(kp.is_causal != 0, kp.varlen_slots != 0 ? kp.b : 0, (kp.varlen_slots != 0 ? (int64_t)1 : (int64_t)kp.b) * kp.h_k * kp.n_split, kp.seqlen_q, kp.seqlen_k, kp.n_k_tiles, (kp.d == 64 && !bwd_use_mfma16(kp, true)) ? 2 : 1, (int64_t)4 * kp.seqlen_q * kp.d * kp.h_ratio / kp.n_split);
    return FA_DISPATCH(launch_dkdv_t, kp, dtype, s);
}

}  // namespace fa
