// fa_bwd.hip — fused attention backward for MI355X (gfx950, CDNA4).
//
// Three kernels on one stream, mirroring the reference's run_flash_bwd
// (csrc/flash_attn/src/flash_bwd_launch_template.h:69-146) but re-designed for wave64 / MFMA:
//
//   fa_bwd_dot_do_o_kernel  D[b,h,i] = sum_d dO*O                 (flash_bwd_preprocess_kernel.h:23-96)
//   fa_bwd_dq_kernel        dQ = scale * sum_j dS_ij K_j           (flash_bwd_kernel.h:28-825)
//   fa_bwd_dkdv_kernel      dV = sum_i P_ij^T dO_i, dK = scale * sum_i dS_ij^T Q_i   (:842-1676)
//
// Like the reference this is the deterministic, atomics-free 7-GEMM form (S and dP are
// recomputed in both kernels).  Differences that matter on CDNA4:
//   * dQ kernel uses the forward's "swapped" layout (lane = query column), so LSE_i and D_i
//     are lane scalars and dS^T feeds the dQ^T MFMA straight from registers;
//   * dK/dV kernel uses the un-swapped layout (lane = key column), K/V fragments stay in
//     registers for the whole Q loop, P and dS feed dV^T / dK^T MFMAs straight from registers,
//     Q^T / dO^T operands come from hardware transposing LDS reads;
//   * the GQA group loop is fused into the dK/dV kernel (the reference materialises per-q-head
//     dK/dV and reduces with torch::sum_out, flash_api.cpp:265-272,301-312).
#include "fa_device.hpp"
#include "fa_params.hpp"

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
    const int batch = blockIdx.z, head = blockIdx.y;
    int sq = p.seqlen_q;
    int64_t row0 = 0, o_boff = (int64_t)batch * p.o.batch, do_boff = (int64_t)batch * p.dout.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int beg = p.cu_seqlens_q[batch];
        sq = p.cu_seqlens_q[batch + 1] - beg;
        row0 = beg;
        o_boff = do_boff = 0;
    }
    const int m0 = blockIdx.x * kDotRowsPerBlock;
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
__global__ __launch_bounds__(kDqThreads, 2) void fa_bwd_dq_kernel(const BwdKernelParams p) {
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int TILEB = kDqBlockN * ROWB;
    constexpr int CPT = (kDqBlockN * SLOTS) / kDqThreads;
    __shared__ __attribute__((aligned(16))) char smem_raw[(4 * TILEB > kDqBlockM * ROWB) ? 4 * TILEB : kDqBlockM * ROWB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint32_t tile, bh;
    decode_block(blockIdx.x, p.n_q_tiles, (uint32_t)(p.b * p.h), tile, bh);
    if (CAUSAL) tile = p.n_q_tiles - 1 - tile;
    const int batch = bh / p.h, head = bh % p.h, head_k = head / p.h_ratio;

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch, v_boff = (int64_t)batch * p.v.batch,
            do_boff = (int64_t)batch * p.dout.batch, dq_boff = (int64_t)batch * p.dq.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int qb = p.cu_seqlens_q[batch], kb = p.cu_seqlens_k[batch];
        sq = p.cu_seqlens_q[batch + 1] - qb;
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
    T* dq_base = uniform_ptr((T*)p.dq_ptr + dq_boff + (q_row0 + m0) * p.dq.row + (int64_t)head * p.dq.head);
    const T* k_base = uniform_ptr((const T*)p.k_ptr + k_boff + k_row0 * p.k.row + (int64_t)head_k * p.k.head);
    const T* v_base = uniform_ptr((const T*)p.v_ptr + v_boff + k_row0 * p.v.row + (int64_t)head_k * p.v.head);
    const int64_t stat_off = ((int64_t)batch * p.h + head) * p.lse_row_stride + m0;

    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), do_rowb = (uint32_t)(p.dout.row * 2), dq_rowb = (uint32_t)(p.dq.row * 2),
                   k_rowb = (uint32_t)(p.k.row * 2), v_rowb = (uint32_t)(p.v.row * 2);
    const rsrc_t q_rs = make_rsrc(q_base, (uint32_t)(rows_here - 1) * q_rowb + ROWB);
    const rsrc_t do_rs = make_rsrc(do_base, (uint32_t)(rows_here - 1) * do_rowb + ROWB);
    const rsrc_t dq_rs = make_rsrc(dq_base, (uint32_t)(rows_here - 1) * dq_rowb + ROWB);
    const rsrc_t k_rs = make_rsrc(k_base, sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const rsrc_t v_rs = make_rsrc(v_base, sk > 0 ? (uint32_t)(sk - 1) * v_rowb + ROWB : 0u);

    int n_tiles = (sk + kDqBlockN - 1) / kDqBlockN;
    if (CAUSAL) {
        const int max_key = m0 + rows_here - 1 + delta;
        n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kDqBlockN + 1);
    }

    const int q_row = wave * 32 + l31;
    const int wave_q_lo = m0 + wave * 32, wave_q_hi = wave_q_lo + 31;

    uint32_t st_goff_k[CPT], st_goff_v[CPT], st_loff[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int chunk = tid + c * kDqThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
        st_goff_k[c] = row * k_rowb + slot * 16;
        st_goff_v[c] = row * v_rowb + slot * 16;
        st_loff[c] = lds_tile_off<D>(row, slot);
    }
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
    float lse2 = 0.f, dsum = 0.f;   // rows past the end keep 0 (they contribute nothing: Q = dO = 0)
    if (q_row < rows_here) {
        lse2 = p.lse_ptr[stat_off + q_row] * kLog2e;
        dsum = p.dsum_ptr[stat_off + q_row];
    }
    const float c = p.scale_log2e;

    f32x16 dqacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[db][r] = 0.f;

    u32x4 st_k[CPT], st_v[CPT];
    if (n_tiles > 0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            st_k[i] = buf_load16(k_rs, st_goff_k[i]);
            st_v[i] = buf_load16(v_rs, st_goff_v[i]);
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            lds_write16(smem, st_loff[i], st_k[i]);
            lds_write16(smem + 2 * TILEB, st_loff[i], st_v[i]);
        }
    }

    for (int t = 0; t < n_tiles; ++t) {
        const int n0 = t * kDqBlockN;
        FA_LDS char* kbuf = smem + (t & 1) * TILEB;
        FA_LDS char* vbuf = smem + 2 * TILEB + (t & 1) * TILEB;
        __syncthreads();
        const bool more = (t + 1 < n_tiles);
        if (more) {
            const uint32_t gk = (uint32_t)(n0 + kDqBlockN) * k_rowb, gv = (uint32_t)(n0 + kDqBlockN) * v_rowb;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                st_k[i] = buf_load16(k_rs, gk + st_goff_k[i]);
                st_v[i] = buf_load16(v_rs, gv + st_goff_v[i]);
            }
        }
        const bool wave_active = !CAUSAL || (n0 <= wave_q_hi + delta);
        if (wave_active) {
            const bool need_mask = (n0 + kDqBlockN > sk) || (CAUSAL && (n0 + kDqBlockN - 1 > wave_q_lo + delta));
            const int lim = CAUSAL ? min(sk - 1, m0 + q_row + delta) : sk - 1;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {           // two 32-key halves, keeps S/dP at 16+16 regs
                f32x16 sacc, dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 kf = lds_read16(kbuf, row_rd[ks] + bi * 32 * ROWB);
                    sacc = LP<T>::mfma(kf, qf[ks], sacc);          // S^T = K Q^T
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 vf = lds_read16(vbuf, row_rd[ks] + bi * 32 * ROWB);
                    dpacc = LP<T>::mfma(vf, dof[ks], dpacc);       // dP^T = V dO^T
                }
                // P = exp(s*scale - LSE) (flash_bwd_kernel.h:474), dS = P * (dP - D) (:490)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pv = fast_exp2(__builtin_fmaf(sacc[r], c, -lse2));
                    if (need_mask) {
                        const int key = n0 + 32 * bi + c_row(r, hi);
                        pv = key <= lim ? pv : 0.f;
                    }
                    sacc[r] = pv * (dpacc[r] - dsum);
                }
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
        if (more) {
            FA_LDS char* kn = smem + ((t + 1) & 1) * TILEB;
            FA_LDS char* vn = smem + 2 * TILEB + ((t + 1) & 1) * TILEB;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                lds_write16(kn, st_loff[i], st_k[i]);
                lds_write16(vn, st_loff[i], st_v[i]);
            }
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
// dK/dV kernel: workgroup = 4 waves = 128 keys of one (batch, kv head); loops over the GQA
// group's query heads and over 64-row Q/dO tiles.
// =============================================================================================
constexpr int kKvThreads = 256;
constexpr int kKvBlockN = 128;   // keys per workgroup (32 per wave)
constexpr int kKvBlockM = 64;    // query rows per staged tile

template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(kKvThreads, 1) void fa_bwd_dkdv_kernel(const BwdKernelParams p) {
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int TILEB = kKvBlockM * ROWB;                 // one Q (or dO) tile
    constexpr int CPT = (kKvBlockM * SLOTS) / kKvThreads;   // 16B chunks / thread / tile
    constexpr int STATB = 2 * kKvBlockM * 4;                // lse2 + dsum of one tile
    constexpr int MAINB = (4 * TILEB > 2 * kKvBlockN * ROWB) ? 4 * TILEB : 2 * kKvBlockN * ROWB;
    // LDS: Q[2] | dO[2] | stats[2]; dK / dV tiles alias Q/dO in the epilogue
    __shared__ __attribute__((aligned(16))) char smem_raw[MAINB + 2 * STATB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* stat = smem + MAINB;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint32_t tile, bhk;
    decode_block(blockIdx.x, p.n_k_tiles, (uint32_t)(p.b * p.h_k), tile, bhk);
    const int batch = bhk / p.h_k, head_k = bhk % p.h_k;

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch, v_boff = (int64_t)batch * p.v.batch,
            do_boff = (int64_t)batch * p.dout.batch, dk_boff = (int64_t)batch * p.dk.batch, dv_boff = (int64_t)batch * p.dv.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int qb = p.cu_seqlens_q[batch], kb = p.cu_seqlens_k[batch];
        sq = p.cu_seqlens_q[batch + 1] - qb;
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
    const uint32_t k_rowb = (uint32_t)(p.k.row * 2), v_rowb = (uint32_t)(p.v.row * 2), dk_rowb = (uint32_t)(p.dk.row * 2),
                   dv_rowb = (uint32_t)(p.dv.row * 2), q_rowb = (uint32_t)(p.q.row * 2), do_rowb = (uint32_t)(p.dout.row * 2);
    const rsrc_t k_rs = make_rsrc(k_base, (uint32_t)(keys_here - 1) * k_rowb + ROWB);
    const rsrc_t v_rs = make_rsrc(v_base, (uint32_t)(keys_here - 1) * v_rowb + ROWB);
    const rsrc_t dk_rs = make_rsrc(dk_base, (uint32_t)(keys_here - 1) * dk_rowb + ROWB);
    const rsrc_t dv_rs = make_rsrc(dv_base, (uint32_t)(keys_here - 1) * dv_rowb + ROWB);

    // Q-tile range: key j is visible to query i iff i >= j - delta
    const int n_q_tiles = (sq + kKvBlockM - 1) / kKvBlockM;
    int qt_begin = 0;
    if (CAUSAL) qt_begin = max(0, n0 - delta) / kKvBlockM;
    const int tiles_per_head = max(0, n_q_tiles - qt_begin);
    const int n_iters = tiles_per_head * p.h_ratio;

    const int key_row = wave * 32 + l31;                 // this lane's key inside the 128-key block
    const int wave_k_lo = n0 + wave * 32, wave_k_hi = wave_k_lo + 31;

    uint32_t st_goff_q[CPT], st_goff_do[CPT], st_loff[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int chunk = tid + c * kKvThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
        st_goff_q[c] = row * q_rowb + slot * 16;
        st_goff_do[c] = row * do_rowb + slot * 16;
        st_loff[c] = lds_tile_off<D>(row, slot);
    }
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

    // B operands held for the whole loop: K^T and V^T fragments of this lane's key
    u32x4 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = buf_load16(k_rs, (uint32_t)key_row * k_rowb + (2 * ks + hi) * 16);
        vf[ks] = buf_load16(v_rs, (uint32_t)key_row * v_rowb + (2 * ks + hi) * 16);
    }
    const float c = p.scale_log2e;

    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[db][r] = 0.f; dvacc[db][r] = 0.f; }

    // iteration -> (query head, q tile) and the global pieces needed to stage it
    auto tile_coords = [&](int it, int& hq, int& m0) {
        const int g = it / tiles_per_head;
        hq = head_k * p.h_ratio + g;
        m0 = (qt_begin + (it - g * tiles_per_head)) * kKvBlockM;
    };
    u32x4 st_q[CPT], st_do[CPT];
    float st_stat = 0.f;
    auto issue_loads = [&](int it) {
        int hq, m0;
        tile_coords(it, hq, m0);
        const int rows = min(kKvBlockM, sq - m0);
        const T* qb = uniform_ptr((const T*)p.q_ptr + q_boff + (q_row0 + m0) * p.q.row + (int64_t)hq * p.q.head);
        const T* dob = uniform_ptr((const T*)p.do_ptr + do_boff + (q_row0 + m0) * p.dout.row + (int64_t)hq * p.dout.head);
        const rsrc_t q_rs = make_rsrc(qb, (uint32_t)(rows - 1) * q_rowb + ROWB);
        const rsrc_t do_rs = make_rsrc(dob, (uint32_t)(rows - 1) * do_rowb + ROWB);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            st_q[i] = buf_load16(q_rs, st_goff_q[i]);
            st_do[i] = buf_load16(do_rs, st_goff_do[i]);
        }
        // threads 0..63 fetch LSE (scaled to log2 units), 64..127 fetch D; rows past the end -> 0
        const int64_t so = ((int64_t)batch * p.h + hq) * p.lse_row_stride + m0;
        st_stat = 0.f;
        if (tid < 2 * kKvBlockM) {
            const int r = tid & (kKvBlockM - 1);
            if (r < rows) st_stat = (tid < kKvBlockM) ? p.lse_ptr[so + r] * kLog2e : p.dsum_ptr[so + r];
        }
    };
    auto land_loads = [&](int buf) {
        FA_LDS char* qd = smem + buf * TILEB;
        FA_LDS char* dd = smem + 2 * TILEB + buf * TILEB;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            lds_write16(qd, st_loff[i], st_q[i]);
            lds_write16(dd, st_loff[i], st_do[i]);
        }
        if (tid < 2 * kKvBlockM) *(FA_LDS float*)(stat + buf * STATB + tid * 4) = st_stat;
    };

    if (n_iters > 0) {
        issue_loads(0);
        land_loads(0);
    }

    for (int it = 0; it < n_iters; ++it) {
        int hq, m0;
        tile_coords(it, hq, m0);
        const int buf = it & 1;
        FA_LDS char* qbuf = smem + buf * TILEB;
        FA_LDS char* dobuf = smem + 2 * TILEB + buf * TILEB;
        FA_LDS char* sbuf = stat + buf * STATB;
        __syncthreads();
        const bool more = (it + 1 < n_iters);
        if (more) issue_loads(it + 1);

        // wave-level causal skip: all 64 rows of the tile are above the diagonal for all 32 keys
        const bool wave_active = !CAUSAL || (wave_k_lo <= m0 + kKvBlockM - 1 + delta);
        if (wave_active) {
            const bool need_mask = CAUSAL && (wave_k_hi > m0 + delta);
            const int key = n0 + key_row;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {           // two 32-row halves of the Q tile
                f32x16 sacc, dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 qa = lds_read16(qbuf, row_rd[ks] + bi * 32 * ROWB);
                    sacc = LP<T>::mfma(qa, kf[ks], sacc);           // S = Q K^T  (rows = queries, lane = key)
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 da = lds_read16(dobuf, row_rd[ks] + bi * 32 * ROWB);
                    dpacc = LP<T>::mfma(da, vf[ks], dpacc);         // dP = dO V^T
                }
                f32x16 pacc;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // registers 4*g4.. = query rows 32*bi + 8*g4 + 4*hi + {0..3}
                    const int rbase = 32 * bi + 8 * g4 + 4 * hi;
                    const f32x4 l4 = *(const FA_LDS f32x4*)(sbuf + rbase * 4);
                    const f32x4 d4 = *(const FA_LDS f32x4*)(sbuf + kKvBlockM * 4 + rbase * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        float pv = fast_exp2(__builtin_fmaf(sacc[r], c, -l4[e]));
                        if (need_mask) pv = (key <= m0 + rbase + e + delta) ? pv : 0.f;
                        pacc[r] = pv;
                        sacc[r] = pv * (dpacc[r] - d4[e]);
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const u32x4 pf = pack_c_half<T>(pacc, half);    // P rounded (flash_bwd_kernel.h:1359)
                    const u32x4 dsf = pack_c_half<T>(sacc, half);   // dS rounded (:1360)
                    const int ts = 2 * bi + half;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const u32x2 a0 = lds_read_tr8(dobuf, tr_rd[0][db] + ts * 16 * ROWB);
                        const u32x2 a1 = lds_read_tr8(dobuf, tr_rd[1][db] + ts * 16 * ROWB);
                        const u32x4 dot = {a0.x, a0.y, a1.x, a1.y};
                        dvacc[db] = LP<T>::mfma(dot, pf, dvacc[db]);       // dV^T += dO^T P
                        const u32x2 b0 = lds_read_tr8(qbuf, tr_rd[0][db] + ts * 16 * ROWB);
                        const u32x2 b1 = lds_read_tr8(qbuf, tr_rd[1][db] + ts * 16 * ROWB);
                        const u32x4 qt = {b0.x, b0.y, b1.x, b1.y};
                        dkacc[db] = LP<T>::mfma(qt, dsf, dkacc[db]);       // dK^T += Q^T dS
                    }
                }
            }
        }
        if (more) land_loads(buf ^ 1);
    }

    // epilogue: dK *= scale (flash_bwd_kernel.h:1652-1654); round; stage; whole-row stores
    __syncthreads();
    FA_LDS char* dk_t = smem;
    FA_LDS char* dv_t = smem + kKvBlockN * ROWB;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            u32x2 w;
            w.x = LP<T>::pack2(dkacc[db][4 * g4 + 0] * p.scale, dkacc[db][4 * g4 + 1] * p.scale);
            w.y = LP<T>::pack2(dkacc[db][4 * g4 + 2] * p.scale, dkacc[db][4 * g4 + 3] * p.scale);
            lds_write8(dk_t, lds_tile_off<D>(key_row, 4 * db + g4) + 8 * hi, w);
            u32x2 x;
            x.x = LP<T>::pack2(dvacc[db][4 * g4 + 0], dvacc[db][4 * g4 + 1]);
            x.y = LP<T>::pack2(dvacc[db][4 * g4 + 2], dvacc[db][4 * g4 + 3]);
            lds_write8(dv_t, lds_tile_off<D>(key_row, 4 * db + g4) + 8 * hi, x);
        }
    __syncthreads();
    constexpr int O_CHUNKS = (kKvBlockN * SLOTS) / kKvThreads;
#pragma unroll
    for (int i = 0; i < O_CHUNKS; ++i) {
        const int chunk = tid + i * kKvThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
        buf_store16(dk_rs, (uint32_t)row * dk_rowb + slot * 16, lds_read16(dk_t, lds_tile_off<D>(row, slot)));
        buf_store16(dv_rs, (uint32_t)row * dv_rowb + slot * 16, lds_read16(dv_t, lds_tile_off<D>(row, slot)));
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int D>
static hipError_t launch_dot_t(const BwdKernelParams& kp, hipStream_t s) {
    dim3 grid((kp.seqlen_q + kDotRowsPerBlock - 1) / kDotRowsPerBlock, kp.h, kp.b);
    hipLaunchKernelGGL((fa_bwd_dot_do_o_kernel<T, D>), grid, dim3(kDotThreads), 0, s, kp);
    return hipGetLastError();
}
template <typename T, int D>
static hipError_t launch_dq_t(const BwdKernelParams& kp, hipStream_t s) {
    const uint32_t grid = kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    if (grid == 0) return hipSuccess;
    if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, true>), dim3(grid), dim3(kDqThreads), 0, s, kp);
    else hipLaunchKernelGGL((fa_bwd_dq_kernel<T, D, false>), dim3(grid), dim3(kDqThreads), 0, s, kp);
    return hipGetLastError();
}
template <typename T, int D>
static hipError_t launch_dkdv_t(const BwdKernelParams& kp, hipStream_t s) {
    const uint32_t grid = kp.n_k_tiles * (uint32_t)kp.b * (uint32_t)kp.h_k;
    if (grid == 0) return hipSuccess;
    if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
    else hipLaunchKernelGGL((fa_bwd_dkdv_kernel<T, D, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
    return hipGetLastError();
}

#define FA_DISPATCH(FN, kp, dtype, s)                                                              \
    ((dtype) == 0 ? ((kp).d == 128 ? FN<_Float16, 128>(kp, s) : FN<_Float16, 64>(kp, s))           \
                  : ((kp).d == 128 ? FN<__bf16, 128>(kp, s) : FN<__bf16, 64>(kp, s)))

hipError_t launch_bwd_dot_do_o(BwdKernelParams kp, int dtype, hipStream_t s) { return FA_DISPATCH(launch_dot_t, kp, dtype, s); }
hipError_t launch_bwd_dq(BwdKernelParams kp, int dtype, hipStream_t s) {
    kp.n_q_tiles = (uint32_t)((kp.seqlen_q + kDqBlockM - 1) / kDqBlockM);
    return FA_DISPATCH(launch_dq_t, kp, dtype, s);
}
hipError_t launch_bwd_dkdv(BwdKernelParams kp, int dtype, hipStream_t s) {
    kp.n_k_tiles = (uint32_t)((kp.seqlen_k + kKvBlockN - 1) / kKvBlockN);
    return FA_DISPATCH(launch_dkdv_t, kp, dtype, s);
}

}  // namespace fa
