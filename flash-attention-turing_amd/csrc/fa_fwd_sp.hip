// fa_fwd_sp.hip — software-pipelined forward kernel ("sp" schedule) for MI355X (gfx950).
//
// Measured on the chip (tools/ubench.hip): a wave that issues ONLY MFMAs starves the VALU of the
// wave it shares a SIMD with down to ONE instruction per MFMA slot (32 cycles), whatever the
// priorities, while a single wave that interleaves 4-5 VALU instructions between its own MFMAs
// runs them for free (38.8 cycles per MFMA+4 VALU; two such waves per SIMD: 37 cycles per MFMA).
// So the softmax must not live in a separate wave/phase from the MFMAs: it has to be woven into
// the MFMA stream of the SAME wave.  The online-softmax data dependence (S -> P -> PV) is
// broken by pipelining across K/V tiles: in iteration u a wave issues
//     region 1:  O^T += V(u-1)^T P(u-1)^T     (16 MFMAs)   ||  row max of S(u)             (VALU)
//     region 2:  S(u+1)^T = K(u+1) Q^T        (16 MFMAs)   ||  P(u) = exp2(S(u) - m), sums (VALU)
// so every MFMA has independent VALU work next to it.  The (deferred, rare) running-max refresh
// is decided between the two regions, when O already contains P.V of tile u-1, so the O rescale
// it implies is exact; P lives in ONE fragment set (written in region 2, consumed in the next
// iteration's region 1) and only S needs two register sets (loop unrolled by two).
//
// Everything else (tiling, fragment layouts, swizzled LDS image, LDS-DMA staging, epilogue) is
// shared with fa_fwd.hip: 8 waves x 32 query rows, 64-key tiles, K/V rings filled by
// hand-issued buffer_load...lds (4-tile rings), ONE workgroup barrier per PAIR of tiles.
#include "fa_device.hpp"
#include "fa_params.hpp"

#include <type_traits>

namespace fa {

constexpr int kSpThreads = 512;
constexpr int kSpBlockM = 256;
constexpr int kSpBlockN = 64;
constexpr float kSpDeferLog2 = 8.0f;   // refresh the running max only when a row outgrows it by 2^8

template <int N>
FA_DEV void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else static_assert(N == 0 || N == 2 || N == 4, "add the literal");
}

template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(kSpThreads, 2) void fa_fwd_sp_kernel(const FwdKernelParams p) {
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int TILEB = kSpBlockN * ROWB;
    constexpr int RING = 4;           // K and V rings: 4 tiles each (two being read, two in flight)
    constexpr int LDSB = (2 * RING * TILEB > kSpBlockM * ROWB) ? 2 * RING * TILEB : kSpBlockM * ROWB;
    __shared__ __attribute__((aligned(16))) char smem_raw[LDSB];     // the ONLY LDS object
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* kring = smem;
    FA_LDS char* vring = smem + RING * TILEB;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint32_t tile, bh;
    decode_block(blockIdx.x, p.n_q_tiles, (uint32_t)(p.b * p.h), tile, bh);
    if (CAUSAL) tile = p.n_q_tiles - 1 - tile;
    const int batch = bh / p.h, head = bh % p.h, head_k = head / p.h_ratio;

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch,
            v_boff = (int64_t)batch * p.v.batch, o_boff = (int64_t)batch * p.o.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int q_beg = p.cu_seqlens_q[batch], k_beg = p.cu_seqlens_k[batch];
        sq = p.cu_seqlens_q[batch + 1] - q_beg;
        sk = p.cu_seqlens_k[batch + 1] - k_beg;
        q_row0 = q_beg; k_row0 = k_beg;
        q_boff = k_boff = v_boff = o_boff = 0;
    }
    const int m0 = tile * kSpBlockM;
    if (m0 >= sq) return;
    const int delta = sk - sq;
    const int rows_here = min(kSpBlockM, sq - m0);

    const T* q_base = uniform_ptr((const T*)p.q_ptr + q_boff + (q_row0 + m0) * p.q.row + (int64_t)head * p.q.head);
    const T* k_base = uniform_ptr((const T*)p.k_ptr + k_boff + k_row0 * p.k.row + (int64_t)head_k * p.k.head);
    const T* v_base = uniform_ptr((const T*)p.v_ptr + v_boff + k_row0 * p.v.row + (int64_t)head_k * p.v.head);
    T* o_base = uniform_ptr((T*)p.o_ptr + o_boff + (q_row0 + m0) * p.o.row + (int64_t)head * p.o.head);
    float* lse_base = p.lse_ptr + ((int64_t)batch * p.h + head) * p.lse_row_stride + m0;
    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), k_rowb = (uint32_t)(p.k.row * 2),
                   v_rowb = (uint32_t)(p.v.row * 2), o_rowb = (uint32_t)(p.o.row * 2);
    const rsrc_t q_rs = make_rsrc(q_base, (uint32_t)(rows_here - 1) * q_rowb + ROWB);
    const rsrc_t o_rs = make_rsrc(o_base, (uint32_t)(rows_here - 1) * o_rowb + ROWB);
    const srd_t k_srd = make_srd(k_base, sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const srd_t v_srd = make_srd(v_base, sk > 0 ? (uint32_t)(sk - 1) * v_rowb + ROWB : 0u);

    int n_tiles = (sk + kSpBlockN - 1) / kSpBlockN;
    if (CAUSAL) {
        const int max_key = m0 + rows_here - 1 + delta;
        n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kSpBlockN + 1);
    }

    const int q_row = wave * 32 + l31;
    const int wave_q_lo = m0 + wave * 32;

    // ---- LDS-DMA staging tables (see fa_fwd.hip) ------------------------------------------------
    constexpr int DPW = SLOTS / 8;
    uint32_t dma_goff_k[DPW], dma_goff_v[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int chunk = (wave * DPW + i) * 64 + lane;
        const int row = chunk / SLOTS, phys = chunk % SLOTS;
        const int slot = lds_tile_logical_slot<D>(row, phys);
        dma_goff_k[i] = row * k_rowb + slot * 16;
        dma_goff_v[i] = row * v_rowb + slot * 16;
    }
    const uint32_t lds_k0 = lds_addr(kring) + (uint32_t)wave * DPW * 1024;
    const uint32_t lds_v0 = lds_addr(vring) + (uint32_t)wave * DPW * 1024;
    auto dma_tile = [&](const srd_t& srd, const uint32_t (&goff)[DPW], uint32_t row0_bytes, uint32_t lds_dst) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma16_to_lds_hidden(srd, row0_bytes + goff[i], lds_dst + i * 1024);
    };

    uint32_t k_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_rd[ks] = lds_tile_off<D>(l31, 2 * ks + hi);
    uint32_t v_rd[2][DB];
    {
        const int L = lane & 15, g = (lane >> 4) & 1;
#pragma unroll
        for (int sec = 0; sec < 2; ++sec)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                v_rd[sec][db] = lds_tile_off<D>(4 * hi + 8 * sec + (L >> 2), 4 * db + 2 * g + ((L & 3) >> 1)) + 8 * (L & 1);
    }

    u32x4 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = buf_load16(q_rs, (uint32_t)q_row * q_rowb + (2 * ks + hi) * 16);

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = kNegBig, l_run = 0.f;
    const float c = p.scale_log2e;

    // ---- prologue: K(0), K(1), K(2), V(0); V slot 3 = zeros ("V(-1)"); past-the-end tiles arrive as zeros ----
    if (n_tiles > 0) {
        dma_tile(k_srd, dma_goff_k, 0u, lds_k0);
        dma_tile(v_srd, dma_goff_v, 0u, lds_v0);
        dma_tile(k_srd, dma_goff_k, (uint32_t)kSpBlockN * k_rowb, lds_k0 + TILEB);
        dma_tile(k_srd, dma_goff_k, (uint32_t)(2 * kSpBlockN) * k_rowb, lds_k0 + 2 * TILEB);
        dma_tile(v_srd, dma_goff_v, 0x80000000u, lds_v0 + 3 * TILEB);     // out of range -> zeros
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    wait_vmcnt<0>();
    __syncthreads();


    // ---- building blocks ----------------------------------------------------------------------
    auto qk_into = [&](f32x16 (&s)[2], int ring_slot) {           // S^T = K(ring_slot) Q^T
        FA_LDS char* kbuf = kring + ring_slot * TILEB;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[bi][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = lds_read16(kbuf, k_rd[ks] + bi * 32 * ROWB);
                s[bi] = LP<T>::mfma(kf, qf[ks], s[bi]);
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400);
            }
        }
    };
    auto pv_from = [&](const u32x4 (&pf)[4], int ring_slot) {   // (pf shadows the kernel-level array on purpose)     // O^T += V(ring_slot)^T P^T
        FA_LDS char* vbuf = vring + ring_slot * TILEB;
#pragma unroll
        for (int ts = 0; ts < 4; ++ts)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const u32x2 a0 = lds_read_tr8(vbuf, v_rd[0][db] + ts * 16 * ROWB);
                const u32x2 a1 = lds_read_tr8(vbuf, v_rd[1][db] + ts * 16 * ROWB);
                const u32x4 vf = {a0.x, a0.y, a1.x, a1.y};
                oacc[db] = LP<T>::mfma(vf, pf[ts], oacc[db]);
                // keep hipcc from hoisting every fragment read of the tile to the top (that costs
                // 60+ VGPRs and spills): LDS reads / MFMAs stay inside their group of MFMAs, plain
                // VALU / SALU / transcendental work may move across freely
                if (db == DB - 1) __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400);
            }
    };
    auto apply_mask = [&](f32x16 (&s)[2], int u) {
        const int n0 = u * kSpBlockN;
        const bool need_mask = (n0 + kSpBlockN > sk) || (CAUSAL && (n0 + kSpBlockN - 1 > wave_q_lo + delta));
        if (need_mask) {
            const int lim = CAUSAL ? min(sk - 1, m0 + q_row + delta) : sk - 1;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = n0 + 32 * bi + c_row(r, hi);
                    s[bi][r] = key <= lim ? s[bi][r] : -INFINITY;
                }
        }
    };
    auto row_max = [&](const f32x16 (&s)[2]) {
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        return max_both_halves(mx);
    };
    auto exp_pack = [&](f32x16 (&s)[2], u32x4 (&pf)[4], float mc) {
        float psum = 0.f;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(__builtin_fmaf(s[bi][r], c, -mc));
                s[bi][r] = pv;
                psum += pv;
            }
        l_run += psum;
#pragma unroll
        for (int ts = 0; ts < 4; ++ts) pf[ts] = pack_c_half<T>(s[ts >> 1], ts & 1);
    };

    // Tiles [0, n_main) need no mask for any row of the workgroup.
    int n_main = min(n_tiles, sk / kSpBlockN);
    if (CAUSAL) n_main = min(n_main, max(0, (m0 + delta + 1) / kSpBlockN));

    // One pipelined iteration for tile u.  s_cur = S(u) (complete), s_next <- S(u+1); pf holds
    // P(u-1) on entry and P(u) on exit.  Ring slot of a tile = tile index mod 4.
    u32x4 pf[4];
    auto iteration = [&](int u, f32x16 (&s_cur)[2], f32x16 (&s_next)[2]) {
        // ---- region 1: P.V of the PREVIOUS tile (MFMA)  ||  row max of THIS tile (VALU) ----
        if (u >= n_main) apply_mask(s_cur, u);        // diagonal / ragged tiles only (wave-uniform)
        const float mx = row_max(s_cur);
        pv_from(pf, (u + 3) & 3);                     // V(u-1)
        // ---- decision: refresh the running max? (wave-uniform, rare after the first tile).
        // O already contains P.V of tile u-1, so it can be rescaled right here. ----
        if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > kSpDeferLog2) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
        // ---- region 2: QK^T of the NEXT tile (MFMA)  ||  exp / sum / pack of THIS tile (VALU) ----
        exp_pack(s_cur, pf, m_run * c);
        qk_into(s_next, (u + 1) & 3);                 // K(u+1)
    };

    f32x16 sA[2], sB[2];
    if (n_tiles > 0) {
        qk_into(sA, 0);                               // pipeline fill: S(0)
        __syncthreads();                              // K(0)'s slot is re-filled by the first pair's DMA
        // Tile 0 has no previous tile: P(-1) = 0 and "V(-1)" (ring slot 3) was zero-filled by the
        // prologue, so its P.V adds exact zeros and every tile runs the SAME two code copies
        // (more copies of the body make hipcc spill).
#pragma unroll
        for (int ts = 0; ts < 4; ++ts) pf[ts] = u32x4{0u, 0u, 0u, 0u};
        // Tiles are processed in pairs (2j, 2j+1) with ONE workgroup barrier per pair.  Pair j reads
        // K(2j+1), K(2j+2), V(2j-1), V(2j); at its start it launches K(2j+3), K(2j+4), V(2j+1),
        // V(2j+2) into the ring slots (tile index mod 4) whose tenants K(2j-1), K(2j), V(2j-3),
        // V(2j-2) were last read in pair j-1; they are waited for (vmcnt(0)) at the end of the
        // pair, a whole pair (~6000 cycles) later, and read in pair j+1.  Always issued (past-the-end
        // rows read as zeros without touching memory).
        for (int u = 0; u < n_tiles; u += 2) {
            dma_tile(k_srd, dma_goff_k, (uint32_t)((u + 3) * kSpBlockN) * k_rowb, lds_k0 + ((u + 3) & 3) * TILEB);
            dma_tile(k_srd, dma_goff_k, (uint32_t)((u + 4) * kSpBlockN) * k_rowb, lds_k0 + ((u + 4) & 3) * TILEB);
            dma_tile(v_srd, dma_goff_v, (uint32_t)((u + 1) * kSpBlockN) * v_rowb, lds_v0 + ((u + 1) & 3) * TILEB);
            dma_tile(v_srd, dma_goff_v, (uint32_t)((u + 2) * kSpBlockN) * v_rowb, lds_v0 + ((u + 2) & 3) * TILEB);
            iteration(u, sA, sB);
            if (u + 1 < n_tiles) iteration(u + 1, sB, sA);
            wait_vmcnt<0>();
            __syncthreads();
        }
        pv_from(pf, (n_tiles - 1) & 3);               // drain: P.V of the last tile
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    const float l_tot = sum_both_halves(l_run);
    const float inv = l_tot > 0.f ? fast_rcp(l_tot) : 0.f;
    const float lse = l_tot > 0.f ? (m_run * c + fast_log2(l_tot)) * kLn2 : 0.f;
    if (hi == 0 && q_row < rows_here) lse_base[q_row] = lse;
    wait_vmcnt<0>();                                  // stray DMA pieces must not land on the O tile
    __syncthreads();
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            u32x2 w;
            w.x = LP<T>::pack2(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
            w.y = LP<T>::pack2(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
            lds_write8(smem, lds_tile_off<D>(q_row, 4 * db + g4) + 8 * hi, w);
        }
    __syncthreads();
    constexpr int O_CHUNKS = (kSpBlockM * SLOTS) / kSpThreads;
#pragma unroll
    for (int i = 0; i < O_CHUNKS; ++i) {
        const int chunk = tid + i * kSpThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
        buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, lds_read16(smem, lds_tile_off<D>(row, slot)));
    }
}

template <typename T, int D>
static hipError_t launch_sp_t(const FwdKernelParams& kp, hipStream_t stream) {
    const uint32_t grid = kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    if (grid == 0) return hipSuccess;
    if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_sp_kernel<T, D, true>), dim3(grid), dim3(kSpThreads), 0, stream, kp);
    else hipLaunchKernelGGL((fa_fwd_sp_kernel<T, D, false>), dim3(grid), dim3(kSpThreads), 0, stream, kp);
    return hipGetLastError();
}

hipError_t launch_fwd_sp(FwdKernelParams kp, int dtype, hipStream_t stream) {
    kp.varlen_slots = 0;   // plain varlen grid
    kp.n_q_tiles = (uint32_t)((kp.seqlen_q + kSpBlockM - 1) / kSpBlockM);
    if (dtype == 0) return kp.d == 128 ? launch_sp_t<_Float16, 128>(kp, stream) : launch_sp_t<_Float16, 64>(kp, stream);
    return kp.d == 128 ? launch_sp_t<__bf16, 128>(kp, stream) : launch_sp_t<__bf16, 64>(kp, stream);
}

}  // namespace fa
