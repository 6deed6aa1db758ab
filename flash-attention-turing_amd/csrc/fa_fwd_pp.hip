// fa_fwd_pp.hip — fused attention forward for MI355X (gfx950, CDNA4): two-group ping-pong schedule.
//
// Replaces the reference's flash_fwd_kernel / compute_attn_1rowblock
// (csrc/flash_attn/src/flash_fwd_kernel.h:23-789, launch flash_fwd_launch_template.h:45-86)
// with a from-scratch wave64 / MFMA 32x32x16 design:
//
//   * workgroup = 8 waves (512 threads) = one 256-row Q tile of one (batch, head);
//     each wave owns 32 query rows for the whole K loop (Q fragments live in VGPRs).
//   * K/V tiles of 64 keys arrive by LDS-DMA (buffer_load ... lds) into 3-deep XOR-swizzled rings.
//   * S^T = K * Q^T is computed ("swapped" QK^T) so that one lane owns one query column of the
//     32x32 accumulator: row max / row sum are in-lane + ONE half-wave exchange
//     (v_permlane32_swap), no LDS, no shuffle trees.
//   * P^T never leaves registers: the S^T accumulator layout is re-used directly as the MFMA B
//     operand of O^T = V^T * P^T, and V^T fragments are fetched with the hardware transposing
//     LDS read (ds_read_b64_tr_b16) using the SAME k-slot permutation.
//   * softmax in base 2: p = exp2(s*c - m*c), c = log2(e)/sqrt(d)  (v_exp_f32 is exp2).
//   * causal (bottom-right aligned, mask.h:172) skips fully masked tiles per workgroup AND per
//     wave; only diagonal / tail tiles take the element mask.
//   * O is normalised, rounded to fp16/bf16, staged through LDS and stored as whole rows.
//   * two wave groups (waves 0-3 / 4-7) run the matrix phase of one against the softmax phase of the other.
//
// Semantics follow SURVEY.md Appendix A; dead rows produce O = 0 and LSE = 0.0
// (flash_fwd_kernel.h:720-728,767-771) without relying on a zero pre-fill of the outputs.
// Earlier alternatives (one-barrier-per-tile baseline, 4-wave variant, single-stream software pipeline) and the
// timing-only ablation switches were removed from the product in round 2; they are in the history at commit a1ce086.
#include "fa_device.hpp"
#include "fa_params.hpp"

#include <type_traits>

// FA_ABL: TIMING-ONLY ablations for tools/ablate_fwd.py (results are WRONG when non-zero; never shipped:
// build.py does not define it).  bit0: no v_exp in the softmax phase; bit1: no fma/exp/row-sum at all;
// bit2: the matrix phase reads only every other K / V fragment from LDS (half the LDS bytes per MFMA);
// bit3: causal diagonal-band tiles run through the unmasked steady-state loop (what if a band tile cost a full tile?);
// bit4: no wave-level causal skip inside the band (every wave computes every band tile);
// bit5: no epilogue output (O staging + stores, LSE) - where does the per-workgroup fixed cost sit?; bit6: no Q load from HBM.

namespace fa {

constexpr int kFwdThreads = 512;
constexpr int kFwdBlockM = 256;
constexpr int kFwdBlockN = 64;

constexpr float kPpDeferLog2 = 6.0f;

template <typename T, int D, bool CAUSAL>
// D = 64: ask for 4 waves per SIMD = TWO 8-wave workgroups per CU (its LDS rings are half the D = 128 size).  The non-causal
// instance fitted 128 registers anyway; the causal one took 140 and silently ran one workgroup per CU.  Forcing 128 costs no
// spill and no extra instruction in any loop; interleaved A/B, causal forward (profiles/r1_fwd_d64_occupancy_ab.log):
// 0.90-0.93x time at 8k, 0.96x at 16k, 0.75x at 2k, 0.71x at 512; non-causal and D = 128 unchanged; outputs bit-identical.
#define FA_PP_MIN_WAVES(D) ((D) == 64 ? 4 : 2)
__global__ __launch_bounds__(kFwdThreads, FA_PP_MIN_WAVES(D)) void fa_fwd_pp_kernel(const FwdKernelParams p) {
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int TILEB = kFwdBlockN * ROWB;
    constexpr int RING = 3;
    constexpr int LDSB = (2 * RING * TILEB > kFwdBlockM * ROWB) ? 2 * RING * TILEB : kFwdBlockM * ROWB;
    __shared__ __attribute__((aligned(16))) char smem_raw[LDSB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* kring = smem;
    FA_LDS char* vring = smem + RING * TILEB;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;

    int tile, batch, head, tiles_seq;
    if (!decode_work<kFwdBlockM>(blockIdx.x, p.n_q_tiles, p.varlen_slots, p.cu_seqlens_q, p.b, p.h, tile, batch, head, tiles_seq)) return;
    if (CAUSAL) tile = tiles_seq - 1 - tile;      // heaviest (latest) query tiles first
    const int head_k = head / p.h_ratio;

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch,
            v_boff = (int64_t)batch * p.v.batch, o_boff = (int64_t)batch * p.o.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int q_beg = p.cu_seqlens_q[batch], k_beg = p.cu_seqlens_k[batch];
        // a sequence longer than the declared max_seqlen_q is clamped: the padded LSE row holds max_seqlen_q entries only
        sq = min(p.cu_seqlens_q[batch + 1] - q_beg, p.seqlen_q);
        sk = p.cu_seqlens_k[batch + 1] - k_beg;
        q_row0 = q_beg; k_row0 = k_beg;
        q_boff = k_boff = v_boff = o_boff = 0;
    }
    const int m0 = tile * kFwdBlockM;
    if (m0 >= sq) return;
    const int delta = sk - sq;
    const int rows_here = min(kFwdBlockM, sq - m0);

    const T* q_base = uniform_ptr((const T*)p.q_ptr + q_boff + (q_row0 + m0) * p.q.row + (int64_t)head * p.q.head);
    const T* k_base = uniform_ptr((const T*)p.k_ptr + k_boff + k_row0 * p.k.row + (int64_t)head_k * p.k.head);
    const T* v_base = uniform_ptr((const T*)p.v_ptr + v_boff + k_row0 * p.v.row + (int64_t)head_k * p.v.head);
    T* o_base = uniform_ptr((T*)p.o_ptr + o_boff + (q_row0 + m0) * p.o.row + (int64_t)head * p.o.head);
    float* lse_base = p.lse_ptr + ((int64_t)batch * p.h + head) * p.lse_row_stride + m0;
    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), k_rowb = (uint32_t)(p.k.row * 2),
                   v_rowb = (uint32_t)(p.v.row * 2), o_rowb = (uint32_t)(p.o.row * 2);
    const rsrc_t q_rs = make_rsrc(q_base, (uint32_t)(rows_here - 1) * q_rowb + ROWB);
    const rsrc_t o_rs = make_rsrc(o_base, (uint32_t)(rows_here - 1) * o_rowb + ROWB);
    const rsrc_t k_rs = make_rsrc(k_base, sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const rsrc_t v_rs = make_rsrc(v_base, sk > 0 ? (uint32_t)(sk - 1) * v_rowb + ROWB : 0u);

    int n_tiles = (sk + kFwdBlockN - 1) / kFwdBlockN;
    if (CAUSAL) {
        const int max_key = m0 + rows_here - 1 + delta;
        n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kFwdBlockN + 1);
    }

    const int q_row = wave * 32 + l31;
    const int wave_q_lo = m0 + wave * 32, wave_q_hi = wave_q_lo + 31;

    // LDS-DMA staging: the tile image in LDS is lane-linear per wave instruction (1 KiB = 64
    // lanes x 16 B), so wave w moves the DPW 1-KiB pieces [w*DPW, (w+1)*DPW) of every tile and
    // the XOR swizzle is applied to the per-lane SOURCE offset.
    constexpr int DPW = SLOTS / 8;            // DMA instructions per wave per tile (2 for d=128, 1 for d=64)
    uint32_t dma_goff_k[DPW], dma_goff_v[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int chunk = (wave * DPW + i) * 64 + lane;          // physical 16-byte chunk inside the tile
        const int row = chunk / SLOTS, phys = chunk % SLOTS;
        const int slot = lds_tile_logical_slot<D>(row, phys);
        dma_goff_k[i] = row * k_rowb + slot * 16;
        dma_goff_v[i] = row * v_rowb + slot * 16;
    }
    const uint32_t dma_loff = (uint32_t)wave * DPW * 1024;       // wave-uniform LDS offset of this wave's pieces
    auto dma_tile = [&](rsrc_t rs, const uint32_t (&goff)[DPW], uint32_t row0_bytes, FA_LDS char* ring_slot) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma16_to_lds(rs, row0_bytes + goff[i], ring_slot + dma_loff + i * 1024);
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

    // ---- prologue: K(0), K(1), V(0) into the rings (past-the-end tiles arrive as zeros) ----------
    if (n_tiles > 0) {
        dma_tile(k_rs, dma_goff_k, 0u, kring);
        dma_tile(v_rs, dma_goff_v, 0u, vring);
        dma_tile(k_rs, dma_goff_k, (uint32_t)kFwdBlockN * k_rowb, kring + TILEB);
    }
    // every wave reads rows DMA-ed by the other waves: own pieces landed (vmcnt), THEN the barrier (the back-off
    // barrier of gfx950 does not imply a vmcnt drain; ROCm 7.2 happens to emit one here, this makes it a guarantee)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (group == 1) __syncthreads();          // group B runs one phase behind group A

    f32x16 sacc[2];
    u32x4 pf[4];
    int ring_u = 0, ring_um1 = 2, ring_up1 = 1;       // u % 3, (u-1) % 3 == (u+2) % 3, (u+1) % 3

    // ---- phase bodies ---------------------------------------------------------------------------
    auto pv_step = [&]() {                            // O^T += V(u-1)^T P(u-1)^T
        FA_LDS char* vbuf = vring + ring_um1 * TILEB;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int ts = 0; ts < 4; ++ts) {
                const u32x2 a0 = lds_read_tr8(vbuf, v_rd[0][db] + ts * 16 * ROWB);
                const u32x2 a1 = lds_read_tr8(vbuf, v_rd[1][db] + ts * 16 * ROWB);
                const u32x4 vf = {a0.x, a0.y, a1.x, a1.y};
                oacc[db] = LP<T>::mfma(vf, pf[ts], oacc[db]);
            }
    };
    auto qk_step = [&]() {                            // S(u)^T = K(u) Q^T
        FA_LDS char* kbuf = kring + ring_u * TILEB;
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
    };
    // K(u+2) and V(u+1) start flying into their ring slots at the START of S(u): the S phase
    // reads no LDS, so hipcc's conservative "LDS-DMA may alias any LDS read" wait lands on the
    // barrier that ends the phase, by which time (the partner's matrix phase is longer than this
    // wave's softmax) the data has arrived.  Previous tenants K(u-1) / V(u-2) were last read in
    // M(u-1) of the other group, at least one barrier ago.
    // The DMA issue cost (~60-100 cycles per 1-KiB piece) is split between the two phases: the V
    // pieces are issued here, at the start of S(u) (compiler-visible builtin, see above); the K
    // pieces are issued by hand at the very END of M(u), after the phase's last LDS read, where the
    // wave would otherwise just wait for its partner at the barrier -- hidden from hipcc so that the
    // barrier ending the matrix phase does not drain them; they are retired by the explicit
    // vmcnt(0) at the end of S(u).  K(u+2)'s slot tenant K(u-1) was last read in M(u-1) of the other
    // group, one barrier before the earliest issue.
    const srd_t k_srd = make_srd(k_base, sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const uint32_t lds_k0 = lds_addr(kring) + dma_loff;
    auto issue_dma_k = [&](int u) {
        if (u + 2 < n_tiles) {
#pragma unroll
            for (int i = 0; i < DPW; ++i)
                dma16_to_lds_hidden(k_srd, (uint32_t)((u + 2) * kFwdBlockN) * k_rowb + dma_goff_k[i], lds_k0 + ring_um1 * TILEB + i * 1024);
        }
    };
    auto issue_dma = [&](int u) {
        if (u + 1 < n_tiles) dma_tile(v_rs, dma_goff_v, (uint32_t)((u + 1) * kFwdBlockN) * v_rowb, vring + ring_up1 * TILEB);
    };
    auto softmax_step = [&](int u, auto masked) {
        const int n0 = u * kFwdBlockN;
        if constexpr (decltype(masked)::value) {
            const bool need_mask = (n0 + kFwdBlockN > sk) || (CAUSAL && (n0 + kFwdBlockN - 1 > wave_q_lo + delta));
            if (need_mask) {
                const int lim = CAUSAL ? min(sk - 1, m0 + q_row + delta) : sk - 1;
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = n0 + 32 * bi + c_row(r, hi);
                        sacc[bi][r] = key <= lim ? sacc[bi][r] : -INFINITY;
                    }
            }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
        mx = max_both_halves(mx);
        // refresh the running max only if some row of the wave outgrew it by > 2^kPpDeferLog2
        if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > kPpDeferLog2) != 0) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
        // (scalar v_fma / v_add on purpose: the float2 form -- v_pk_fma_f32 / v_pk_add_f32, half the
        // instructions -- measured 5 % SLOWER next to the partner wave's MFMAs, tools/fwd_ab.py)
        const float mc = m_run * c;
        float psum = 0.f;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(__builtin_fmaf(sacc[bi][r], c, -mc));
                psum += pv;
                sacc[bi][r] = pv;
            }
        l_run += psum;
#pragma unroll
        for (int ts = 0; ts < 4; ++ts) pf[ts] = pack_c_half<T>(sacc[ts >> 1], ts & 1);
    };
    auto advance_ring = [&]() {
        ring_um1 = ring_u;
        ring_u = ring_up1;
        ring_up1 = ring_up1 == 2 ? 0 : ring_up1 + 1;
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    // Tiles [0, n_main) are fully visible to every row of the workgroup and fully inside the
    // sequence: no mask, no per-wave skipping -> a branch-free steady-state loop.  The remaining
    // (diagonal / ragged) tiles and the pipeline fill / drain go through the generic body.
    int n_main = min(n_tiles, sk / kFwdBlockN);
    if (CAUSAL) n_main = min(n_main, max(0, (m0 + delta + 1) / kFwdBlockN));

    bool prev_active = false;                         // does this wave hold a P tile whose PV is pending?
    // every S phase ends with: this wave's LDS-DMA pieces have landed (vmcnt) -> workgroup barrier
    auto end_s_phase = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto generic_iter = [&](int u) {
        const bool in_range = u < n_tiles;
        const bool active = in_range && (!CAUSAL || (u * kFwdBlockN <= wave_q_hi + delta));
        if (prev_active) pv_step();
        if (active) qk_step();
        if (!in_range) return;                        // drain iteration: only the pending PV
        issue_dma_k(u);
        __syncthreads();
        issue_dma(u);
        if (active) softmax_step(u, yes{});
        prev_active = active;
        end_s_phase();
        advance_ring();
    };

    int u = 0;
    if (n_main > 0) {                                 // pipeline fill: tile 0 has no pending PV
        qk_step();
        issue_dma_k(0);
        __syncthreads();
        issue_dma(0);
        softmax_step(0, no{});
        prev_active = true;
        end_s_phase();
        advance_ring();
        for (u = 1; u < n_main; ++u) {                // steady state
            pv_step();
            qk_step();
            issue_dma_k(u);
            __syncthreads();
            issue_dma(u);
            softmax_step(u, no{});
            end_s_phase();
            advance_ring();
        }
    }
    for (; u <= n_tiles; ++u) generic_iter(u);        // diagonal / ragged tiles, then the drain
    if (group == 0) __syncthreads();          // group A waits for B's last phase (equal barrier counts)

    // ---- epilogue -----------------------------------------------------------------------------------
    const float l_tot = sum_both_halves(l_run);
    const float inv = l_tot > 0.f ? fast_rcp(l_tot) : 0.f;
    const float lse = l_tot > 0.f ? (m_run * c + fast_log2(l_tot)) * kLn2 : 0.f;
    if (hi == 0 && q_row < rows_here) lse_base[q_row] = lse;
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
    constexpr int O_CHUNKS = (kFwdBlockM * SLOTS) / kFwdThreads;
#pragma unroll
    for (int i = 0; i < O_CHUNKS; ++i) {
        const int chunk = tid + i * kFwdThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
        buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, lds_read16(smem, lds_tile_off<D>(row, slot)));
    }
}


template <typename T, int D>
static hipError_t launch_pp_t(const FwdKernelParams& kp, hipStream_t stream) {
    const uint32_t grid = kp.varlen_slots != 0 ? kp.varlen_slots * (uint32_t)kp.h : kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    if (grid == 0) return hipSuccess;
    if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, true>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
    else hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, false>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
    return hipGetLastError();
}

const char* fwd_kernel_name(int) { return "fa_fwd_pp_kernel"; }

hipError_t launch_fwd(FwdKernelParams kp, int dtype, hipStream_t stream) {
    kp.n_q_tiles = (uint32_t)((kp.seqlen_q + kFwdBlockM - 1) / kFwdBlockM);
    kp.varlen_slots = kp.cu_seqlens_q != nullptr ? varlen_slot_count(kp.total_q, kp.b, kFwdBlockM, kp.n_q_tiles) : 0u;
    if (dtype == 0) return kp.d == 128 ? launch_pp_t<_Float16, 128>(kp, stream) : launch_pp_t<_Float16, 64>(kp, stream);
    return kp.d == 128 ? launch_pp_t<__bf16, 128>(kp, stream) : launch_pp_t<__bf16, 64>(kp, stream);
}

}  // namespace fa
