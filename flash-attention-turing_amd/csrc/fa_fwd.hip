// fa_fwd.hip — fused attention forward for MI355X (gfx950, CDNA4): baseline schedule + dispatch.
// (The shipped default schedule lives in fa_fwd_pp.hip; design notes common to all three are here.)
//
// Replaces the reference's flash_fwd_kernel / compute_attn_1rowblock
// (csrc/flash_attn/src/flash_fwd_kernel.h:23-789, launch flash_fwd_launch_template.h:45-86)
// with a from-scratch wave64 / MFMA 32x32x16 design:
//
//   * workgroup = 8 waves (512 threads) = one 256-row Q tile of one (batch, head);
//     each wave owns 32 query rows for the whole K loop (Q fragments live in VGPRs).
//   * K/V tiles of 64 keys are staged HBM -> VGPR -> LDS (double buffered, XOR-swizzled,
//     one barrier per tile; the next tile's global loads are issued before the MFMAs of the
//     current tile and written to LDS after them).
//   * S^T = K * Q^T is computed ("swapped" QK^T) so that one lane owns one query column of the
//     32x32 accumulator: row max / row sum are in-lane + ONE half-wave exchange
//     (v_permlane32_swap), no LDS, no shuffles trees.
//   * P^T never leaves registers: the S^T accumulator layout is re-used directly as the MFMA B
//     operand of O^T = V^T * P^T, and V^T fragments are fetched with the hardware transposing
//     LDS read (ds_read_b64_tr_b16) using the SAME k-slot permutation.
//   * softmax in base 2: p = exp2(s*c - m*c), c = log2(e)/sqrt(d)  (v_exp_f32 is exp2).
//   * causal (bottom-right aligned, mask.h:172) skips fully masked tiles per workgroup AND per
//     wave; only diagonal / tail tiles take the element mask.
//   * O is normalised, rounded to fp16/bf16, staged through LDS and stored as whole rows.
//
// Semantics follow SURVEY.md Appendix A; dead rows produce O = 0 and LSE = 0.0
// (flash_fwd_kernel.h:720-728,767-771) without relying on a zero pre-fill of the outputs.
#include "fa_device.hpp"
#include "fa_params.hpp"

#include <stdlib.h>
#include <string.h>


namespace fa {

constexpr int kFwdBlockN = 64;   // keys per staged tile (threads and query rows per workgroup follow from WAVES, see the kernel)

// WAVES = 8: the original 512-thread / 256-row workgroup, one per CU.  WAVES = 4 ("simple4"): 256 threads / 128 rows, TWO workgroups
// per CU (64 KB LDS and <= 256 registers each): the two independent workgroups on a CU decorrelate on their own, which is what
// short sequences need (few, short workgroups whose prologue / epilogue are otherwise fully exposed); it doubles the L2 -> LDS
// traffic per unit of work, so it is a candidate for short sequences only.
template <typename T, int D, bool CAUSAL, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void fa_fwd_kernel(const FwdKernelParams p) {
    constexpr int kFwdThreadsW = 64 * WAVES;    // threads per workgroup
    constexpr int kFwdBlockMW = 32 * WAVES;     // query rows per workgroup (32 per wave)
    constexpr int KS = D / 16;          // k-steps of the QK^T contraction
    constexpr int DB = D / 32;          // 32-wide d blocks of the O^T accumulator
    constexpr int ROWB = D * 2;         // bytes per staged row
    constexpr int SLOTS = D / 8;        // 16-byte slots per row
    constexpr int TILEB = kFwdBlockN * ROWB;             // bytes per K (or V) tile
    constexpr int CHUNKS_PER_THREAD = (kFwdBlockN * SLOTS) / kFwdThreadsW;  // 16B chunks / thread / tile
    static_assert(CHUNKS_PER_THREAD >= 1, "tile too small for this many threads");

    // LDS: K[2] | V[2]; the O tile (256 x D) aliases the whole region in the epilogue.
    __shared__ __attribute__((aligned(16))) char smem_raw[(4 * TILEB > kFwdBlockMW * ROWB) ? 4 * TILEB : kFwdBlockMW * ROWB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int l31 = lane & 31;

    // ---- which tile ----------------------------------------------------------------------
    uint32_t tile, bh;
    decode_block(blockIdx.x, p.n_q_tiles, (uint32_t)(p.b * p.h), tile, bh);
    if (CAUSAL) tile = p.n_q_tiles - 1 - tile;  // heaviest (latest) query tiles first
    const int batch = bh / p.h;
    const int head = bh % p.h;
    const int head_k = head / p.h_ratio;

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;      // first row of this sequence in the (packed) tensors
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch,
            v_boff = (int64_t)batch * p.v.batch, o_boff = (int64_t)batch * p.o.batch;
    if (p.cu_seqlens_q != nullptr) {     // varlen (block_info.h:6-21)
        const int q_beg = p.cu_seqlens_q[batch], k_beg = p.cu_seqlens_k[batch];
        sq = p.cu_seqlens_q[batch + 1] - q_beg;
        sk = p.cu_seqlens_k[batch + 1] - k_beg;
        q_row0 = q_beg;
        k_row0 = k_beg;
        q_boff = k_boff = v_boff = o_boff = 0;
    }
    const int m0 = tile * kFwdBlockMW;
    if (m0 >= sq) return;                // tile beyond this sequence (flash_fwd_kernel.h:55-57)

    const int delta = sk - sq;           // causal: key j visible to query i iff j <= i + delta
    const int rows_here = min(kFwdBlockMW, sq - m0);

    const T* q_base = uniform_ptr((const T*)p.q_ptr + q_boff + (q_row0 + m0) * p.q.row + (int64_t)head * p.q.head);
    const T* k_base = uniform_ptr((const T*)p.k_ptr + k_boff + k_row0 * p.k.row + (int64_t)head_k * p.k.head);
    const T* v_base = uniform_ptr((const T*)p.v_ptr + v_boff + k_row0 * p.v.row + (int64_t)head_k * p.v.head);
    T* o_base = uniform_ptr((T*)p.o_ptr + o_boff + (q_row0 + m0) * p.o.row + (int64_t)head * p.o.head);
    float* lse_base = p.lse_ptr + ((int64_t)batch * p.h + head) * p.lse_row_stride + m0;

    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), k_rowb = (uint32_t)(p.k.row * 2),
                   v_rowb = (uint32_t)(p.v.row * 2), o_rowb = (uint32_t)(p.o.row * 2);
    // Valid byte extents from each base: rows [0, n) of a head slice. Anything past is
    // out of range for the SRD => loads return 0, stores are dropped.
    const rsrc_t q_rs = make_rsrc(q_base, (uint32_t)(rows_here - 1) * q_rowb + ROWB);
    const rsrc_t o_rs = make_rsrc(o_base, (uint32_t)(rows_here - 1) * o_rowb + ROWB);
    const rsrc_t k_rs = make_rsrc(k_base, sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const rsrc_t v_rs = make_rsrc(v_base, sk > 0 ? (uint32_t)(sk - 1) * v_rowb + ROWB : 0u);

    // ---- K-tile range ----------------------------------------------------------------------
    int n_tiles = (sk + kFwdBlockN - 1) / kFwdBlockN;
    if (CAUSAL) {
        const int max_key = m0 + rows_here - 1 + delta;   // last key any row of this tile can see
        n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kFwdBlockN + 1);
    }

    // ---- per-lane constants ------------------------------------------------------------------
    const int q_row = wave * 32 + l31;                    // row inside the 256-row tile
    const int wave_q_lo = m0 + wave * 32;                 // first / last global query row of this wave
    const int wave_q_hi = wave_q_lo + 31;

    // staging: thread -> (row, slot) of the 16-byte chunks it moves for every K and V tile
    uint32_t st_goff_k[CHUNKS_PER_THREAD], st_goff_v[CHUNKS_PER_THREAD], st_loff[CHUNKS_PER_THREAD];
#pragma unroll
    for (int c = 0; c < CHUNKS_PER_THREAD; ++c) {
        const int chunk = tid + c * kFwdThreadsW;
        const int row = chunk / SLOTS, slot = chunk % SLOTS;
        st_goff_k[c] = row * k_rowb + slot * 16;
        st_goff_v[c] = row * v_rowb + slot * 16;
        st_loff[c] = lds_tile_off<D>(row, slot);
    }
    // K row-read offsets: A fragment of S^T block bi, k-step ks = K[32*bi + l31][16*ks + 8*hi ..+7]
    uint32_t k_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_rd[ks] = lds_tile_off<D>(l31, 2 * ks + hi);
    // V transposed-read offsets: A fragment of O^T block db, k-slice t (16 keys), in two
    // 4-key reads (sec): rows 16*t + 4*hi + 8*sec + (L>>2), cols 32*db + 16*g + 4*(L&3)..
    uint32_t v_rd[2][DB];
    {
        const int L = lane & 15, g = (lane >> 4) & 1;
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                v_rd[sec][db] = lds_tile_off<D>(4 * hi + 8 * sec + (L >> 2), 4 * db + 2 * g + ((L & 3) >> 1)) + 8 * (L & 1);
    }

    // ---- prologue: Q fragments (B operand of S^T): Q[q_row][16*ks + 8*hi ..+7] -------------
    u32x4 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = buf_load16(q_rs, (uint32_t)q_row * q_rowb + (2 * ks + hi) * 16);

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = kNegBig;   // running max of raw (unscaled) scores of this lane's query row
    float l_run = 0.f;       // this half-wave's partial row sum (the two halves are added at the end)
    const float c = p.scale_log2e;

    u32x4 st_k[CHUNKS_PER_THREAD], st_v[CHUNKS_PER_THREAD];
    if (n_tiles > 0) {
#pragma unroll
        for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
            st_k[i] = buf_load16(k_rs, st_goff_k[i]);
            st_v[i] = buf_load16(v_rs, st_goff_v[i]);
        }
#pragma unroll
        for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
            lds_write16(smem, st_loff[i], st_k[i]);
            lds_write16(smem + 2 * TILEB, st_loff[i], st_v[i]);
        }
    }

    for (int t = 0; t < n_tiles; ++t) {
        const int n0 = t * kFwdBlockN;
        FA_LDS char* kbuf = smem + (t & 1) * TILEB;
        FA_LDS char* vbuf = smem + 2 * TILEB + (t & 1) * TILEB;
        __syncthreads();  // tile t is in LDS; everyone is done reading the other buffer (tile t-1)

        const bool more = (t + 1 < n_tiles);
        if (more) {       // issue next tile's HBM loads now, park them in VGPRs under the MFMAs
            const uint32_t gk = (uint32_t)(n0 + kFwdBlockN) * k_rowb, gv = (uint32_t)(n0 + kFwdBlockN) * v_rowb;
#pragma unroll
            for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
                st_k[i] = buf_load16(k_rs, gk + st_goff_k[i]);
                st_v[i] = buf_load16(v_rs, gv + st_goff_v[i]);
            }
        }

        // wave-level causal skip: every key of this tile is masked for all 32 rows of the wave
        const bool wave_active = !CAUSAL || (n0 <= wave_q_hi + delta);
        if (wave_active) {
            // ---- S^T (64 keys x 32 queries) = K_tile * Q^T --------------------------------
            f32x16 sacc[2];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[bi][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 kf = lds_read16(kbuf, k_rd[ks] + bi * 32 * ROWB);
                    sacc[bi] = LP<T>::mfma(kf, qf[ks], sacc[bi]);
                }
            }
            // ---- mask (diagonal / tail tiles only; wave-uniform branch) -------------------
            const bool need_mask = (n0 + kFwdBlockN > sk) || (CAUSAL && (n0 + kFwdBlockN - 1 > wave_q_lo + delta));
            if (need_mask) {
                const int lim = CAUSAL ? min(sk - 1, m0 + q_row + delta) : sk - 1;  // last visible key of this row
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = n0 + 32 * bi + c_row(r, hi);
                        sacc[bi][r] = key <= lim ? sacc[bi][r] : -INFINITY;
                    }
            }
            // ---- online softmax, one query row per lane pair (lane, lane^32) ---------------
            float mx = sacc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
            mx = max_both_halves(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2((m_run - m_new) * c);
            const float mc = m_new * c;
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(__builtin_fmaf(sacc[bi][r], c, -mc));
                    sacc[bi][r] = pv;
                    psum += pv;
                }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            // P^T as MFMA B fragments (rounded to fp16/bf16 like the reference, :485,670)
            u32x4 pf[4];
#pragma unroll
            for (int ts = 0; ts < 4; ++ts) pf[ts] = pack_c_half<T>(sacc[ts >> 1], ts & 1);
            // ---- O^T (D x 32 queries) += V_tile^T * P^T -----------------------------------
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int ts = 0; ts < 4; ++ts) {
                    const u32x2 a0 = lds_read_tr8(vbuf, v_rd[0][db] + ts * 16 * ROWB);
                    const u32x2 a1 = lds_read_tr8(vbuf, v_rd[1][db] + ts * 16 * ROWB);
                    const u32x4 vf = {a0.x, a0.y, a1.x, a1.y};
                    oacc[db] = LP<T>::mfma(vf, pf[ts], oacc[db]);
                }
        }

        if (more) {       // land the prefetched tile in the other LDS buffer
            FA_LDS char* kn = smem + ((t + 1) & 1) * TILEB;
            FA_LDS char* vn = smem + 2 * TILEB + ((t + 1) & 1) * TILEB;
#pragma unroll
            for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
                lds_write16(kn, st_loff[i], st_k[i]);
                lds_write16(vn, st_loff[i], st_v[i]);
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    const float l_tot = sum_both_halves(l_run);
    const float inv = l_tot > 0.f ? fast_rcp(l_tot) : 0.f;   // dead row -> O = 0 (flash_fwd_kernel.h:720-728)
    // LSE = m*scale + ln(l) (flash_fwd_kernel.h:770); 0.0 for dead rows (:767-771)
    const float lse = l_tot > 0.f ? (m_run * c + fast_log2(l_tot)) * kLn2 : 0.f;
    if (hi == 0 && q_row < rows_here) lse_base[q_row] = lse;

    __syncthreads();  // all waves finished reading K/V buffers -> reuse LDS for the O tile
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // registers 4*g4..4*g4+3 = 4 consecutive d: 32*db + 8*g4 + 4*hi + {0..3}
            u32x2 w;
            w.x = LP<T>::pack2(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
            w.y = LP<T>::pack2(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
            lds_write8(smem, lds_tile_off<D>(q_row, 4 * db + g4) + 8 * hi, w);
        }
    __syncthreads();
    constexpr int O_CHUNKS = (kFwdBlockMW * SLOTS) / kFwdThreadsW;
#pragma unroll
    for (int i = 0; i < O_CHUNKS; ++i) {
        const int chunk = tid + i * kFwdThreadsW;
        const int row = chunk / SLOTS, slot = chunk % SLOTS;
        const u32x4 val = lds_read16(smem, lds_tile_off<D>(row, slot));
        buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, val);   // rows >= rows_here fall outside the SRD
    }
}

// --------------------------------------------------------------------------------------------
// Forward schedules (all parity-tested, tests/test_attention_gpu.py::fwd_impl):
//   pp     (default) two-group ping-pong, fa_fwd_pp.hip      -- fastest measured (tools/fwd_ab.py)
//   sp     software-pipelined single stream, fa_fwd_sp.hip   -- within ~6 % of pp
//   simple one barrier per tile, no overlap, this file       -- bisecting aid / readable baseline
// FA_FWD_IMPL=simple|sp|pp selects one at load time; fa_debug_set_fwd_impl() switches inside a
// process for interleaved A/B timing.
static int fwd_impl() {
    static int impl = [] {
        const char* e = getenv("FA_FWD_IMPL");
        if (e != nullptr && strcmp(e, "simple") == 0) return 0;
        if (e != nullptr && strcmp(e, "sp") == 0) return 2;
        if (e != nullptr && strcmp(e, "simple4") == 0) return 3;
        return 1;
    }();
    return impl;
}
static int g_fwd_impl_override = -1;
void set_fwd_impl(int impl) { g_fwd_impl_override = impl; }

template <typename T, int D, int WAVES>
static hipError_t launch_fwd_t(const FwdKernelParams& kp, hipStream_t stream) {
    const uint32_t grid = kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    if (grid == 0) return hipSuccess;
    if (kp.is_causal)
        hipLaunchKernelGGL((fa_fwd_kernel<T, D, true, WAVES>), dim3(grid), dim3(64 * WAVES), 0, stream, kp);
    else
        hipLaunchKernelGGL((fa_fwd_kernel<T, D, false, WAVES>), dim3(grid), dim3(64 * WAVES), 0, stream, kp);
    return hipGetLastError();
}
template <int WAVES>
static hipError_t launch_fwd_w(FwdKernelParams kp, int dtype, hipStream_t stream) {
    kp.varlen_slots = 0;   // bisecting schedules keep the plain varlen grid
    kp.n_q_tiles = (uint32_t)((kp.seqlen_q + 32 * WAVES - 1) / (32 * WAVES));
    if (dtype == 0) {
        return kp.d == 128 ? launch_fwd_t<_Float16, 128, WAVES>(kp, stream) : launch_fwd_t<_Float16, 64, WAVES>(kp, stream);
    } else {
        return kp.d == 128 ? launch_fwd_t<__bf16, 128, WAVES>(kp, stream) : launch_fwd_t<__bf16, 64, WAVES>(kp, stream);
    }
}

hipError_t launch_fwd(FwdKernelParams kp, int dtype, hipStream_t stream) {
    const int impl = g_fwd_impl_override >= 0 ? g_fwd_impl_override : fwd_impl();
    if (impl == 1) return launch_fwd_pp(kp, dtype, stream);
    if (impl == 2) return launch_fwd_sp(kp, dtype, stream);
    if (impl == 3) return launch_fwd_w<4>(kp, dtype, stream);
    return launch_fwd_w<8>(kp, dtype, stream);
}

}  // namespace fa
