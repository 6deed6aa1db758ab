// fa_bwd_dq16.hip -- the dQ kernel of fa_bwd.hip (same workgroup shape, K / V double buffer, hand-issued LDS-DMA, D computed in the prologue;
// read its header first) re-tiled for v_mfma_f32_16x16x32_{f16,bf16}, head_dim 128 only.
//
// Why: the chip runs these kernels against its power cap, and the 16x16x32 shape draws less per FLOP (tools/powerbench; the forward's
// second kernel, fa_fwd_pp16.hip, has the measurements).  The backward kernels carry less softmax work per MFMA than the forward, so more
// of that difference should arrive.
//
// Layout (that of fa_fwd_pp16.hip): a lane is (k-group g = lane >> 4, column n = lane & 15) and owns TWO query rows of its wave's 32
// (n and n + 16), so LSE_i and D_i are lane scalars; S^T = K Q^T and dP^T = V dO^T come in 16 x 16 blocks [key block][query column] with
// the contraction in 4 steps of 32 d; a lane's dS values of key blocks 2c and 2c + 1 ARE the 8 k-slots of the B operand of
// dQ^T += K^T dS^T (chunks of 32 keys); every LDS fragment feeds two MFMAs (one per query column).  The two k-slot permutations keep the
// unchanged XOR-swizzled tile image free of bank conflicts (fa_fwd_pp16.hip).
#include <type_traits>

#include "fa_device.hpp"
#include "fa_params.hpp"

namespace fa {

constexpr int kDq16Threads = 512;
constexpr int kDq16BlockM = 256;
// keys per tile: 64 at head_dim 128; head_dim 64 (round 5) also takes 128-key tiles (the same 16 KiB tile images, half the barriers per key)
#ifndef FA_DQ16_D64_BN
#define FA_DQ16_D64_BN 128
#endif
#ifndef FA_DQ16_ABL
#define FA_DQ16_ABL 0         // timing-only ablations (results may be WRONG): 1 = the LDS-DMA of the next tile is not waited for
#endif
#ifndef FA_DQ16_STAGGER_DMA
#define FA_DQ16_STAGGER_DMA 1
#endif
// (Round 6, measured and not kept: FA_DQ16_SKEW - waves 4-7 with their workgroup barrier in FRONT of the tile's last dQ phase, so that the two waves of a SIMD stay a phase
// apart; three-slot K / V rings, loop unrolled x3, no spill in the loops: bit-identical, +1.7..+10 %, profiles/r6_dq_skew_ab.log.)

// FA_DQ16_MIN_WAVES: waves per SIMD the register budget is cut for.  head_dim 64 (round 5): 2 = one workgroup per compute unit on up to 256 registers,
// 4 = two co-resident workgroups on 128 (what the 32x32x16 head_dim-64 kernel runs with).
#ifndef FA_DQ16_MIN_WAVES
#define FA_DQ16_MIN_WAVES(D) 2
#endif

template <typename T, int D, bool CAUSAL, int BN>
__global__ __launch_bounds__(kDq16Threads, FA_DQ16_MIN_WAVES(D)) void fa_bwd_dq16_kernel(const BwdKernelParams p) {
    static_assert(D == 128 || D == 64, "head_dim");
    constexpr int KS = D / 32, DB = D / 16, ROWB = D * 2, SLOTS = D / 8;
    constexpr int kDq16BlockN = BN;
    constexpr int NC = kDq16BlockN / 32;                                 // 32-key chunks (two 16-key score blocks each) per tile
    constexpr int TILEB = kDq16BlockN * ROWB;
    __shared__ __attribute__((aligned(16))) char smem_raw[(4 * TILEB > kDq16BlockM * ROWB) ? 4 * TILEB : kDq16BlockM * ROWB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi_g = (0x2130 >> (4 * g)) & 3;             // d chunks  {0, 3, 1, 2}
    const int pi2_g = (0x3120 >> (4 * g)) & 3;            // key sub-blocks {0, 2, 1, 3}

    int tile, batch, head, tiles_seq;
    if (!decode_work<kDq16BlockM>(blockIdx.x, p.n_q_tiles, p.varlen_slots, p.cu_seqlens_q, p.b, p.h, tile, batch, head, tiles_seq, p.group_heads)) return;
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
    const int m0 = tile * kDq16BlockM;
    if (m0 >= sq) return;
    const int delta = sk - sq;
    const int rows_here = min(kDq16BlockM, sq - m0);

    const T* q_base = uniform_ptr((const T*)p.q_ptr + q_boff + (q_row0 + m0) * p.q.row + (int64_t)head * p.q.head);
    const T* do_base = uniform_ptr((const T*)p.do_ptr + do_boff + (q_row0 + m0) * p.dout.row + (int64_t)head * p.dout.head);
    const T* o_base = uniform_ptr((const T*)p.o_ptr + (p.cu_seqlens_q != nullptr ? 0 : (int64_t)batch * p.o.batch) + (q_row0 + m0) * p.o.row + (int64_t)head * p.o.head);
    T* dq_base = uniform_ptr((T*)p.dq_ptr + dq_boff + (q_row0 + m0) * p.dq.row + (int64_t)head * p.dq.head);
    const T* k_base = uniform_ptr((const T*)p.k_ptr + k_boff + k_row0 * p.k.row + (int64_t)head_k * p.k.head);
    const T* v_base = uniform_ptr((const T*)p.v_ptr + v_boff + k_row0 * p.v.row + (int64_t)head_k * p.v.head);
    const int64_t stat_off = ((int64_t)batch * p.h + head) * p.lse_row_stride + m0;

    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), do_rowb = (uint32_t)(p.dout.row * 2), dq_rowb = (uint32_t)(p.dq.row * 2),
                   k_rowb = (uint32_t)(p.k.row * 2), v_rowb = (uint32_t)(p.v.row * 2), o_rowb = (uint32_t)(p.o.row * 2);
    const rsrc_t q_rs = make_rsrc(q_base, (uint32_t)(rows_here - 1) * q_rowb + ROWB);
    const rsrc_t do_rs = make_rsrc(do_base, (uint32_t)(rows_here - 1) * do_rowb + ROWB);
    const rsrc_t dq_rs = make_rsrc(dq_base, (uint32_t)(rows_here - 1) * dq_rowb + ROWB);
    const rsrc_t o_rs = make_rsrc(o_base, (uint32_t)(rows_here - 1) * o_rowb + ROWB);
    const srd_t k_srd = make_srd(k_base, sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const srd_t v_srd = make_srd(v_base, sk > 0 ? (uint32_t)(sk - 1) * v_rowb + ROWB : 0u);

    int n_tiles = (sk + kDq16BlockN - 1) / kDq16BlockN;
    if (CAUSAL) {
        const int max_key = m0 + rows_here - 1 + delta;
        n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kDq16BlockN + 1);
    }
    const int q_row_a = wave * 32 + n16;                  // this lane's two query rows inside the block: q_row_a, q_row_a + 16
    const int wave_q_lo = m0 + wave * 32, wave_q_hi = wave_q_lo + 31;

    // LDS-DMA staging (hand-issued, fa_device.hpp:dma16_to_lds_hidden): wave w moves the DPW 1-KiB pieces [w*DPW, (w+1)*DPW) of every K
    // and V tile; the swizzle is applied to the source offset.  Buffers: K0 K1 V0 V1.
    constexpr int DPW = BN * SLOTS / 512;
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
    auto dma_tiles = [&](int t, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(k_srd, (uint32_t)(t * kDq16BlockN) * k_rowb + dma_goff_k[i], lds_k0 + buf * TILEB + i * 1024);
            dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(v_srd, (uint32_t)(t * kDq16BlockN) * v_rowb + dma_goff_v[i], lds_v0 + buf * TILEB + i * 1024);
        }
    };
    // row reads of K (A of S^T = K Q^T) and V (A of dP^T = V dO^T): key block kb, k-step ks -> row 16*kb + 4*kPi2[i >> 2] + (i & 3) for
    // A row i = n16, 16-byte slot 4*ks + kPi[g]
    uint32_t row_rd[KS];
    {
        const int row = 4 * ((0x3120 >> (4 * (n16 >> 2))) & 3) + (n16 & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) row_rd[ks] = lds_tile_off<D>(row, 4 * ks + pi_g);
    }
    // transposed reads of K (A of dQ^T += K^T dS^T): lane group g points at the 4 key rows 4*kPi2[g] .. +3 of a 16-key block, 16 d wide
    uint32_t tr_rd[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) tr_rd[db] = lds_tile_off<D>(4 * pi2_g + (n16 >> 2), 2 * db + ((n16 & 3) >> 1)) + 8 * (n16 & 1);

    // B operands held for the whole loop: Q^T and dO^T fragments of this lane's two query rows
    u32x4 qf[KS][2], dof[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            qf[ks][qb] = buf_load16(q_rs, (uint32_t)(q_row_a + 16 * qb) * q_rowb + (4 * ks + pi_g) * 16);
            dof[ks][qb] = buf_load16(do_rs, (uint32_t)(q_row_a + 16 * qb) * do_rowb + (4 * ks + pi_g) * 16);
        }
    // D_i = rowsum(dO_i * O_i) (flash_bwd_preprocess_kernel.h:23-96) from the dO fragments this lane holds anyway: a lane has 4 of the 16
    // 16-byte slots of each of its rows, the four lane groups together all of them.  fp32 products and sums.
    float lse2[2] = {0.f, 0.f}, dsum[2] = {0.f, 0.f};      // rows past the end keep 0 (they contribute nothing: Q = dO = 0)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 of = buf_load16(o_rs, (uint32_t)(q_row_a + 16 * qb) * o_rowb + (4 * ks + pi_g) * 16);      // rows past the end read zeros
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                part += LP<T>::to_float((uint16_t)(of[w] & 0xffff)) * LP<T>::to_float((uint16_t)(dof[ks][qb][w] & 0xffff));
                part += LP<T>::to_float((uint16_t)(of[w] >> 16)) * LP<T>::to_float((uint16_t)(dof[ks][qb][w] >> 16));
            }
        }
        dsum[qb] = sum_four_groups(part);
        const int row = q_row_a + 16 * qb;
        if (row < rows_here) {
            lse2[qb] = p.lse_ptr[stat_off + row] * kLog2e;
            if (g == 0) p.dsum_ptr[stat_off + row] = dsum[qb];
        }
    }
    const float c = p.scale_log2e;
    // loop-invariant blocks of -D_i: every dP chain starts from them, so dP - D costs no VALU
    f32x4 negd[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) negd[qb] = f32x4{-dsum[qb], -dsum[qb], -dsum[qb], -dsum[qb]};

    f32x4 dqacc[DB][2];                                   // dQ^T: d rows 16*db + 4*g + r, query column qb
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) dqacc[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (n_tiles > 0) dma_tiles(0, 0);
    // the Q / dO / LSE / D loads above are compiler-visible: force them home so the hand-counted vmcnt(0) below also covers the DMA pieces
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks][0]), "+v"(qf[ks][1]), "+v"(dof[ks][0]), "+v"(dof[ks][1]));
    asm volatile("" : "+v"(lse2[0]), "+v"(lse2[1]), "+v"(negd[0]), "+v"(negd[1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    auto tile_body = [&](int t, const int BUF) __attribute__((always_inline)) {
        const int n0 = t * kDq16BlockN;
        FA_LDS char* kbuf = smem + BUF * TILEB;
        FA_LDS char* vbuf = smem + 2 * TILEB + BUF * TILEB;
        __syncthreads();      // tile t is in LDS (every wave waited for its pieces); the other buffer is free again
        const bool wave_active = !CAUSAL || (n0 <= wave_q_hi + delta);
        const bool dma_late = FA_DQ16_STAGGER_DMA && wave >= 4 && wave_active;       // (fa_bwd.hip: waves 4-7 issue after their first half)
        if (t + 1 < n_tiles && !dma_late) dma_tiles(t + 1, BUF ^ 1);
        if (wave_active) {
            const bool need_mask = (n0 + kDq16BlockN > sk) || (CAUSAL && (n0 + kDq16BlockN - 1 > wave_q_lo + delta));
            int lim_loc[2];                           // key of element (kb, r) = n0 + 16*kb + 4*kPi2[g] + r: everything lane- or tile-dependent folded once
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                lim_loc[qb] = (CAUSAL ? min(sk - 1, m0 + q_row_a + 16 * qb + delta) : sk - 1) - n0 - 4 * pi2_g;
#pragma unroll
            for (int cch = 0; cch < NC; ++cch) {      // two 32-key halves: S / dP stay at 16 + 16 registers
                f32x4 sacc[2][2], dpacc[2][2];        // [key block of the half][query column]
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const u32x4 kf = lds_read16(kbuf, row_rd[ks] + (2 * cch + kk) * 16 * ROWB);
                        const u32x4 vf = lds_read16(vbuf, row_rd[ks] + (2 * cch + kk) * 16 * ROWB);
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb) {
                            sacc[kk][qb] = LP<T>::mfma16(kf, qf[ks][qb], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sacc[kk][qb]);       // S^T = K Q^T
                            dpacc[kk][qb] = LP<T>::mfma16(vf, dof[ks][qb], ks == 0 ? negd[qb] : dpacc[kk][qb]);                     // dP^T - D = V dO^T - D
                        }
                    }
                }
                // P = exp(s*scale - LSE) (flash_bwd_kernel.h:474), dS = P * (dP - D) (:490)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[kk][qb][r] = fast_exp2(__builtin_fmaf(sacc[kk][qb][r], c, -lse2[qb]));
                if (need_mask) {                                    // wave-uniform branch: interior tiles skip the 2 VALU per element
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) sacc[kk][qb][r] = (16 * (2 * cch + kk) + r) <= lim_loc[qb] ? sacc[kk][qb][r] : 0.f;
                }
                u32x4 dsf[2];                                       // dS^T rounded like the reference (:512), as the B operand of this 32-key chunk
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    dsf[qb].x = LP<T>::pack2(sacc[0][qb][0] * dpacc[0][qb][0], sacc[0][qb][1] * dpacc[0][qb][1]);
                    dsf[qb].y = LP<T>::pack2(sacc[0][qb][2] * dpacc[0][qb][2], sacc[0][qb][3] * dpacc[0][qb][3]);
                    dsf[qb].z = LP<T>::pack2(sacc[1][qb][0] * dpacc[1][qb][0], sacc[1][qb][1] * dpacc[1][qb][1]);
                    dsf[qb].w = LP<T>::pack2(sacc[1][qb][2] * dpacc[1][qb][2], sacc[1][qb][3] * dpacc[1][qb][3]);
                }
                if (cch == 0 && dma_late && t + 1 < n_tiles) dma_tiles(t + 1, BUF ^ 1);
                // dQ^T (16-d blocks x 16 queries) += K^T (16 d x 32 keys) * dS^T (32 keys x 16 queries)
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const u32x2 a0 = lds_read_tr8(kbuf, tr_rd[db] + (32 * cch) * ROWB);
                    const u32x2 a1 = lds_read_tr8(kbuf, tr_rd[db] + (32 * cch + 16) * ROWB);
                    const u32x4 ktf = {a0.x, a0.y, a1.x, a1.y};
                    dqacc[db][0] = LP<T>::mfma16(ktf, dsf[0], dqacc[db][0]);
                    dqacc[db][1] = LP<T>::mfma16(ktf, dsf[1], dqacc[db][1]);
                }
            }
        }
#if !(FA_DQ16_ABL & 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t+1 have landed
#endif
    };
    {
        int t = 0;
        for (; t + 2 <= n_tiles; t += 2) {        // two tiles per trip: the ring slot is a compile-time constant, fragment reads are base + immediate
            tile_body(t, 0);
            tile_body(t + 1, 1);
        }
        if (t < n_tiles) tile_body(t, 0);
    }

    // epilogue: dQ *= scale (flash_bwd_kernel.h:765), round, stage through LDS, whole-row stores
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            u32x2 w;
            w.x = LP<T>::pack2(dqacc[db][qb][0] * p.scale, dqacc[db][qb][1] * p.scale);
            w.y = LP<T>::pack2(dqacc[db][qb][2] * p.scale, dqacc[db][qb][3] * p.scale);
            lds_write8(smem, lds_tile_off<D>(q_row_a + 16 * qb, 2 * db + (g >> 1)) + 8 * (g & 1), w);      // d = 16*db + 4*g + {0..3}
        }
    __syncthreads();
    constexpr int O_CHUNKS = (kDq16BlockM * SLOTS) / kDq16Threads;
#pragma unroll
    for (int i = 0; i < O_CHUNKS; ++i) {
        const int chunk = tid + i * kDq16Threads, row = chunk / SLOTS, slot = chunk % SLOTS;
        buf_store16(dq_rs, (uint32_t)row * dq_rowb + slot * 16, lds_read16(smem, lds_tile_off<D>(row, slot)));
    }
}

hipError_t launch_bwd_dq16(const BwdKernelParams& kp, int dtype, hipStream_t s) {
    const uint32_t grid = kp.varlen_slots != 0 ? kp.varlen_slots * (uint32_t)kp.h : kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    if (grid == 0) return hipSuccess;
    if (kp.d == 64) {      // round 5: the same kernel at head_dim 64 (8 KiB tiles, 2 k-steps, 4 d blocks)
        if (dtype == 0) {
            if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dq16_kernel<_Float16, 64, true, FA_DQ16_D64_BN>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
            else hipLaunchKernelGGL((fa_bwd_dq16_kernel<_Float16, 64, false, FA_DQ16_D64_BN>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
        } else {
            if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dq16_kernel<__bf16, 64, true, FA_DQ16_D64_BN>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
            else hipLaunchKernelGGL((fa_bwd_dq16_kernel<__bf16, 64, false, FA_DQ16_D64_BN>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
        }
        return hipGetLastError();
    }
    if (dtype == 0) {
        if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dq16_kernel<_Float16, 128, true, 64>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
        else hipLaunchKernelGGL((fa_bwd_dq16_kernel<_Float16, 128, false, 64>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
    } else {
        if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dq16_kernel<__bf16, 128, true, 64>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
        else hipLaunchKernelGGL((fa_bwd_dq16_kernel<__bf16, 128, false, 64>), dim3(grid), dim3(kDq16Threads), 0, s, kp);
    }
    return hipGetLastError();
}

}  // namespace fa
