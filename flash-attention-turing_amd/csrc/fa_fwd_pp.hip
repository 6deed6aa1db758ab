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
//   * steady state: the running max is only refreshed when a row outgrows it by 2^6, and the softmax first runs an optimistic
//     pass WITHOUT the row-max reduction (exact path only if a lane's partial row sum exceeds 2^6); the matrix phase is one
//     hand-pipelined stream of MFMAs with LDS fragments requested 4 steps ahead; at D = 128 the loop runs three tiles per trip
//     so that ring slots are compile-time constants (no address arithmetic left in it).
//
// head_dim 128 has a second kernel with the same schedule tiled for v_mfma_f32_16x16x32 (fa_fwd_pp16.hip); launch_fwd at the end of this file
// picks per launch (fa_set_kernel_policy).
//
// Semantics follow SURVEY.md Appendix A; dead rows produce O = 0 and LSE = 0.0
// (flash_fwd_kernel.h:720-728,767-771) without relying on a zero pre-fill of the outputs.
// Earlier alternatives (one-barrier-per-tile baseline, 4-wave variant, single-stream software pipeline) and the
// timing-only ablation switches were removed from the product in round 2; they are in the history at commit a1ce086.
#include <atomic>

#include "fa_device.hpp"
#include "fa_params.hpp"

#include <type_traits>


namespace fa {

constexpr int kFwdThreads = 512;
constexpr int kFwdBlockM = 256;

constexpr float kPpDeferLog2 = 6.0f;

// D = 64: ask for 4 waves per SIMD = TWO 8-wave workgroups per CU (its LDS is half the D = 128 size).  The non-causal
// instance fitted 128 registers anyway; the causal one took 140 and silently ran one workgroup per CU.  Forcing 128 costs no
// spill and no extra instruction in any loop; interleaved A/B, causal forward (profiles/r1_fwd_d64_occupancy_ab.log):
// 0.90-0.93x time at 8k, 0.96x at 16k, 0.75x at 2k, 0.71x at 512; non-causal and D = 128 unchanged; outputs bit-identical.
#define FA_PP_MIN_WAVES(D, BN) ((D) == 64 && (BN) == 64 ? 4 : 2)
// Keys per tile.  D = 128: 64.  D = 64: two shapes, picked by the launcher (the reference has a d = 64 tile of its own as well,
// flash_fwd_launch_template.h:102-111):
//   64 keys  - two workgroups per CU on 128 registers (round 2): one's prologue / epilogue hides behind the other's loop; better for
//              every non-causal length (1-8 %) and for short causal ones;
//   128 keys - the D = 128 structure (16 KiB tiles, 32 MFMAs per matrix phase, one workgroup per CU, x3-unrolled loop, optimistic
//              softmax also under the mask): the per-tile costs that do not shrink with D (two barriers, DMA pieces at ~110 cycles
//              each, the phase hand-over) weigh half as much; 5-9 % faster under a causal mask from 2k keys on in round 3, only from
//              16k keys on since the causal dispatch order of round 4 (fa_params.hpp:causal_group_heads) took the imbalance away.
// Measured ladder 256 .. 16k, both shapes, causal or not: profiles/r3_fwd_d64_tile_ab.log.  -DFA_FWD_D64_BN=64 / 128 pins one shape (A/B).
#ifndef FA_PP_DMA_FUSED
#define FA_PP_DMA_FUSED 1      // (round 5) a wave's two pieces of a K (or V) tile as one asm statement (fa_device.hpp:dma16x2_to_lds_hidden), as in fa_fwd_pp16.hip
#endif
#ifndef FA_FWD_D64_BN
#define FA_FWD_D64_BN 0
#endif
constexpr int kFwdD64WideMinKeys = 16384;     // (round 3: 2048.  With the round-4 causal dispatch order the 64-key shape is ahead up to 8k: profiles/r4_fwd_d64_shapes_after_reorder_ab.log)
// Matrix phase = NPV P*V steps, then NQK QK^T steps.  P*V step j -> (output block db = j % DB, key sub-tile ts = j / DB): consecutive
// MFMAs go to different accumulators (per accumulator the ts order, hence the result, is unchanged; 0.3-0.5 % over db-major at D = 128;
// alternating P*V and QK^T steps measured the same, profiles/r2_fwd_step_order_ab.log).
#define FA_STEP_IS_PV(j) ((j) < NPV)
#define FA_STEP_IDX(j) ((j) < NPV ? (j) : (j) - NPV)
#define FA_PV_DB(j) ((j) % DB)
#define FA_PV_TS(j) ((j) / DB)
#ifndef FA_PP_OPTIMISTIC
#define FA_PP_OPTIMISTIC(D, CAUSAL, BN) ((D) == 128 || (BN) == 128 || !(CAUSAL))      // D = 64 causal with 64-key tiles: the extra path does not fit its 128 registers
#endif

// One workgroup = one 256-row query tile of one (batch, head).  Two multi-item variants were built and measured in round 2 and are
// NOT here (the persistent one is in the history at 1c7aadc..3e5f845, disabled): a persistent grid walking a static, XCD-aware item list, and per-workgroup
// pairs of query tiles (T-1-j, j) for causal balance, both running the K / V rings, the phase offset and the barrier cadence
// straight through item boundaries (next item's first K / V tiles prefetched by the last iterations, its Q block parked in LDS,
// boundary matrix phase = [Q' -> registers; pending P*V; epilogue; first QK^T]).  Both were bit-identical to this kernel and both
// were SLOWER: hipcc gives the tile loop a worse register allocation once it sits inside an item loop (persistent: kernel arguments
// of the next head stay live across it, loop descriptors get spilled, LDS reads serialise: 1.27-1.7x; pairs: 256 VGPRs + 30
// spilled, 1.02-1.05x at 4k-16k, 1.26x at causal 1k, and half the grid at 512).  profiles/r2_fwd_pair_mode_ab.log.
template <typename T, int D, bool CAUSAL, int BN>
__global__ __launch_bounds__(kFwdThreads, FA_PP_MIN_WAVES(D, BN)) void fa_fwd_pp_kernel(const FwdKernelParams p) {
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int kFwdBlockN = BN, NB = BN / 32, NTS = BN / 16;      // keys per tile; 32-key score blocks / 16-key P fragments per tile
    constexpr int TILEB = kFwdBlockN * ROWB;
    constexpr int RING = 3;
    constexpr int RINGB = 2 * RING * TILEB, STAGEB = kFwdBlockM * ROWB;     // D = 128: 96 KiB + 64 KiB = all 160 KiB of the CU
    __shared__ __attribute__((aligned(16))) char smem_raw[RINGB + STAGEB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* kring = smem;
    FA_LDS char* vring = smem + RING * TILEB;
    FA_LDS char* stage = smem + RINGB;            // O block on its way out; every wave touches only its own 32 rows

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;
    const int q_row = wave * 32 + l31;
    const float c = p.scale_log2e;
    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), k_rowb = (uint32_t)(p.k.row * 2),
                   v_rowb = (uint32_t)(p.v.row * 2), o_rowb = (uint32_t)(p.o.row * 2);

    // ---- items of this workgroup ------------------------------------------------------------------------
    int tile, batch, head, tiles_seq;
    if (!decode_work<kFwdBlockM>(blockIdx.x, p.n_q_tiles, p.varlen_slots, p.cu_seqlens_q, p.b, p.h, tile, batch, head, tiles_seq, p.group_heads)) return;
    if (CAUSAL) tile = tiles_seq - 1 - tile;            // heaviest (latest) query tiles first

    // ---- geometry (wave-uniform) -----------------------------------------------------------------------
    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    const bool varlen = p.cu_seqlens_q != nullptr;
    if (varlen) {
        const int q_beg = p.cu_seqlens_q[batch], k_beg = p.cu_seqlens_k[batch];
        // a sequence longer than the declared max_seqlen_q is clamped: the padded LSE row holds max_seqlen_q entries only
        sq = min(p.cu_seqlens_q[batch + 1] - q_beg, p.seqlen_q);
        sk = p.cu_seqlens_k[batch + 1] - k_beg;
        q_row0 = q_beg; k_row0 = k_beg;
    }
    if (tile * kFwdBlockM >= sq) return;
    // row 0 of this (batch, head) in Q / O / LSE: everything an item needs beyond these is its tile index
    const char* q_bh = (const char*)uniform_ptr((const T*)p.q_ptr + (varlen ? 0 : (int64_t)batch * p.q.batch) + q_row0 * p.q.row + (int64_t)head * p.q.head);
    char* o_bh = (char*)uniform_ptr((T*)p.o_ptr + (varlen ? 0 : (int64_t)batch * p.o.batch) + q_row0 * p.o.row + (int64_t)head * p.o.head);
    float* lse_bh = uniform_ptr(p.lse_ptr + ((int64_t)batch * p.h + head) * p.lse_row_stride);
    const int head_k = head / p.h_ratio;
    const srd_t k_srd = make_srd(uniform_ptr((const T*)p.k_ptr + (varlen ? 0 : (int64_t)batch * p.k.batch) + k_row0 * p.k.row + (int64_t)head_k * p.k.head),
                                 sk > 0 ? (uint32_t)(sk - 1) * k_rowb + ROWB : 0u);
    const srd_t v_srd = make_srd(uniform_ptr((const T*)p.v_ptr + (varlen ? 0 : (int64_t)batch * p.v.batch) + k_row0 * p.v.row + (int64_t)head_k * p.v.head),
                                 sk > 0 ? (uint32_t)(sk - 1) * v_rowb + ROWB : 0u);
    auto q_ptr_of = [&](int t) __attribute__((always_inline)) { return (const T*)(q_bh + (uint32_t)(t * kFwdBlockM) * q_rowb); };
    auto o_ptr_of = [&](int t) __attribute__((always_inline)) { return (T*)(o_bh + (uint32_t)(t * kFwdBlockM) * o_rowb); };
    auto rows_of = [&](int t) __attribute__((always_inline)) { return min(kFwdBlockM, sq - t * kFwdBlockM); };
    int m0 = 0, delta = 0, n_tiles = 0, n_main = 0;   // of the CURRENT item
    auto set_current = [&](int t) __attribute__((always_inline)) {
        m0 = t * kFwdBlockM;
        delta = sk - sq;
        n_tiles = (sk + kFwdBlockN - 1) / kFwdBlockN;
        if (CAUSAL) {
            const int max_key = m0 + rows_of(t) - 1 + delta;
            n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kFwdBlockN + 1);
        }
        // Tiles [0, n_main) are fully visible to every row of the workgroup and fully inside the sequence: no mask, no per-wave
        // skipping -> a branch-free steady-state loop.  The remaining (diagonal / ragged) tiles go through the generic body.
        n_main = min(n_tiles, sk / kFwdBlockN);
        if (CAUSAL) n_main = min(n_main, max(0, (m0 + delta + 1) / kFwdBlockN));
    };

    // ---- lane constants --------------------------------------------------------------------------------
    // LDS-DMA staging: the tile image in LDS is lane-linear per wave instruction (1 KiB = 64 lanes x 16 B), so wave w moves
    // the DPW 1-KiB pieces [w*DPW, (w+1)*DPW) of every K / V tile and the XOR swizzle is applied to the per-lane SOURCE offset.
    // Every LDS-DMA of this kernel is issued from inline asm (fa_device.hpp:dma16_to_lds_hidden): hipcc never sees one, so it never
    // parks a vmcnt(0) in front of an LDS read; completion is the explicit vmcnt(0) that ends every softmax phase.
    constexpr int DPW = BN * SLOTS / 512;     // DMA instructions (1 KiB pieces) per wave per tile: 2 (1 for D = 64 with 64-key tiles)
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
    const uint32_t lds_k0 = lds_addr(kring) + dma_loff, lds_v0 = lds_addr(vring) + dma_loff;
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

    set_current(tile);
    if (n_tiles == 0) {
        // a block of rows that sees no key at all (causal with seqlen_q > seqlen_k, or an empty key sequence): O = 0, LSE = 0 (flash_fwd_kernel.h:718,767), written straight from
        // registers - no Q load, no K / V tile, no LDS, no barrier (round 4: such a workgroup took ~38 us through the full prologue and epilogue, profiles/r4_fwd_dead_rows_ab.log)
        const int rows_here = rows_of(tile);
        const rsrc_t o_rs = make_rsrc(uniform_ptr(o_ptr_of(tile)), (uint32_t)(rows_here - 1) * o_rowb + ROWB);
        constexpr int O_CHUNKS_DEAD = (32 * SLOTS) / 64;
#pragma unroll
        for (int i = 0; i < O_CHUNKS_DEAD; ++i) {
            const int chunk = lane + i * 64, row = wave * 32 + chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, u32x4{0u, 0u, 0u, 0u});      // rows >= rows_here fall outside the SRD
        }
        if (lane < 32 && wave * 32 + lane < rows_here) lse_bh[tile * kFwdBlockM + wave * 32 + lane] = 0.f;
        return;
    }

    u32x4 qf[KS];
    {
        const rsrc_t q_rs = make_rsrc(uniform_ptr(q_ptr_of(tile)), (uint32_t)(rows_of(tile) - 1) * q_rowb + ROWB);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = buf_load16(q_rs, (uint32_t)q_row * q_rowb + (2 * ks + hi) * 16);
    }

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = kNegBig, l_run = 0.f;

    int ring_u = 0, ring_um1 = 2, ring_up1 = 1;       // slot of tile u, of u-1 (== u+2), of u+1 in the 3-deep rings
    auto dma_k_tile = [&](const srd_t& srd, int t, int slot) __attribute__((always_inline)) {
#if FA_PP_DMA_FUSED
        if constexpr (DPW == 2) { dma16x2_to_lds_hidden(srd, (uint32_t)(t * kFwdBlockN) * k_rowb, dma_goff_k[0], dma_goff_k[1], lds_k0 + slot * TILEB); return; }
#endif
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma16_to_lds_hidden<false>(srd, (uint32_t)(t * kFwdBlockN) * k_rowb + dma_goff_k[i], lds_k0 + slot * TILEB + i * 1024);
    };
    auto dma_v_tile = [&](const srd_t& srd, int t, int slot) __attribute__((always_inline)) {
#if FA_PP_DMA_FUSED
        if constexpr (DPW == 2) { dma16x2_to_lds_hidden(srd, (uint32_t)(t * kFwdBlockN) * v_rowb, dma_goff_v[0], dma_goff_v[1], lds_v0 + slot * TILEB); return; }
#endif
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma16_to_lds_hidden<false>(srd, (uint32_t)(t * kFwdBlockN) * v_rowb + dma_goff_v[i], lds_v0 + slot * TILEB + i * 1024);
    };

    // ---- prologue: K(0), V(0), K(1) into the rings (past-the-end tiles arrive as zeros) ---------------
    if (n_tiles > 0) {
        dma_k_tile(k_srd, 0, 0);
        dma_v_tile(v_srd, 0, 0);
        dma_k_tile(k_srd, 1, 1);
    }
    // every wave reads rows DMA-ed by the other waves: own pieces landed (vmcnt), THEN the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (group == 1) __syncthreads();          // group B runs one phase behind group A, for the whole life of the workgroup
    // (a static s_setprio 1 / 2 for this younger group changes nothing: +-0.0 % at 4k-16k, profiles/r3_fwd_prio_ab.log)

    f32x16 sacc[NB];
    u32x4 pf[NTS];
    int wave_q_lo = m0 + wave * 32, wave_q_hi = wave_q_lo + 31;

    // ---- phase bodies ---------------------------------------------------------------------------
    auto pv_step = [&]() __attribute__((always_inline)) {                            // O^T += V(u-1)^T P(u-1)^T
        FA_LDS char* vbuf = vring + ring_um1 * TILEB;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int ts = 0; ts < NTS; ++ts) {
                const u32x2 a0 = lds_read_tr8(vbuf, v_rd[0][db] + ts * 16 * ROWB);
                const u32x2 a1 = lds_read_tr8(vbuf, v_rd[1][db] + ts * 16 * ROWB);
                const u32x4 vf = {a0.x, a0.y, a1.x, a1.y};
                oacc[db] = LP<T>::mfma(vf, pf[ts], oacc[db]);
            }
    };
    auto qk_step = [&]() __attribute__((always_inline)) {                            // S(u)^T = K(u) Q^T
        FA_LDS char* kbuf = kring + ring_u * TILEB;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[bi][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = lds_read16(kbuf, k_rd[ks] + bi * 32 * ROWB);
                sacc[bi] = LP<T>::mfma(kf, qf[ks], sacc[bi]);
            }
        }
    };
    // Steady-state matrix phase as ONE software-pipelined stream: 4*DB P*V MFMAs (V(u-1), tile slot ring_um1) then 2*KS QK^T
    // MFMAs (K(u), slot ring_u); the LDS fragment of step j + PF is requested before MFMA j, and the first PF fragments are
    // requested at the END of the wave's previous softmax phase (`pre`), i.e. before the barrier that opens this phase, so the
    // phase starts on an MFMA instead of on an LDS round trip.  Measured with per-phase s_memtime stamps (tools/phase_timing.py,
    // profiles/r2_fwd_phase_timing.log): the period of the ping-pong is the SUM of the two groups' matrix phases (the softmax
    // phases hide behind them), and a matrix phase took 1270 cycles for 1024 cycles of MFMA issue.
    constexpr bool kOptimistic = FA_PP_OPTIMISTIC(D, CAUSAL, BN);
    constexpr int NPV = NTS * DB, NQK = NB * KS, NST = NPV + NQK, PF = (D == 64 && BN == 64) ? 2 : 4;      // fragments in flight (3 / 6 / 8 measured within 0.5 % of 4); D = 64 with 64-key tiles has 128 VGPRs only
    auto m_frag = [&](int j, int slot_v, int slot_k) __attribute__((always_inline)) -> u32x4 {
        if (FA_STEP_IS_PV(j)) {
            const int pj = FA_STEP_IDX(j), db = FA_PV_DB(pj), ts = FA_PV_TS(pj);
            FA_LDS char* vbuf = vring + slot_v * TILEB;
            const u32x2 a0 = lds_read_tr8(vbuf, v_rd[0][db] + ts * 16 * ROWB);
            const u32x2 a1 = lds_read_tr8(vbuf, v_rd[1][db] + ts * 16 * ROWB);
            return u32x4{a0.x, a0.y, a1.x, a1.y};
        }
        const int i = FA_STEP_IDX(j), ks = i / NB, bi = i % NB;
        return lds_read16(kring + slot_k * TILEB, k_rd[ks] + bi * 32 * ROWB);
    };
    // read bases with the ring origin folded in and hidden from the compiler: with compile-time ring slots every fragment read is
    // base + 16-bit immediate (slot * 16 KiB + row block), no address arithmetic left in the loop
    uint32_t k_abs[KS], v_abs[2][DB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { k_abs[ks] = lds_addr(kring) + k_rd[ks]; asm volatile("" : "+v"(k_abs[ks])); }
#pragma unroll
    for (int sec = 0; sec < 2; ++sec)
#pragma unroll
        for (int db = 0; db < DB; ++db) { v_abs[sec][db] = lds_addr(vring) + v_rd[sec][db]; asm volatile("" : "+v"(v_abs[sec][db])); }
    auto m_frag_c = [&](int j, int slot_v, int slot_k) __attribute__((always_inline)) -> u32x4 {
        if (FA_STEP_IS_PV(j)) {
            const int pj = FA_STEP_IDX(j), db = FA_PV_DB(pj), ts = FA_PV_TS(pj);
            const u32x2 a0 = lds_read_tr8((const FA_LDS char*)(uintptr_t)v_abs[0][db], slot_v * TILEB + ts * 16 * ROWB);
            const u32x2 a1 = lds_read_tr8((const FA_LDS char*)(uintptr_t)v_abs[1][db], slot_v * TILEB + ts * 16 * ROWB);
            return u32x4{a0.x, a0.y, a1.x, a1.y};
        }
        const int i = FA_STEP_IDX(j), ks = i / NB, bi = i % NB;
        return lds_read16((const FA_LDS char*)(uintptr_t)k_abs[ks], slot_k * TILEB + bi * 32 * ROWB);
    };
    u32x4 pre[PF];
    auto m_prefetch = [&](int slot_v, int slot_k) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PF; ++j) pre[j] = m_frag(j, slot_v, slot_k);
    };
    auto m_phase = [&]() __attribute__((always_inline)) {
        u32x4 fr[NST];
        static_for<0, PF>([&](auto jc) { fr[decltype(jc)::value] = pre[decltype(jc)::value]; });
        static_for<0, NST>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + PF < NST) fr[j + PF] = m_frag(j + PF, ring_um1, ring_u);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FA_STEP_IS_PV(j)) {
                constexpr int pj = FA_STEP_IDX(j);
                oacc[FA_PV_DB(pj)] = LP<T>::mfma(fr[j], pf[FA_PV_TS(pj)], oacc[FA_PV_DB(pj)]);
            } else {
                constexpr int i = FA_STEP_IDX(j), ks = i / NB, bi = i % NB;
                if constexpr (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[bi][r] = 0.f;
                }
                sacc[bi] = LP<T>::mfma(fr[j], qf[ks], sacc[bi]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // K(u+2) starts flying at the very END of M(u), after the phase's last LDS read, where the wave would otherwise just wait
    // for its partner at the barrier; V(u+1) at the START of S(u), which reads no LDS.  Previous tenants of the slots, K(u-1) and
    // V(u-2), were last read in M(u-1) of the other group, at least one barrier before the earliest issue.  Both are retired by
    // the explicit vmcnt(0) that ends S(u).
    auto issue_dma_k = [&](int u) __attribute__((always_inline)) {
        if (u + 2 < n_tiles) dma_k_tile(k_srd, u + 2, ring_um1);
    };
    auto issue_dma_v = [&](int u) __attribute__((always_inline)) {
        if (u + 1 < n_tiles) dma_v_tile(v_srd, u + 1, ring_up1);
    };
    auto softmax_step = [&](int u, auto masked) __attribute__((always_inline)) {
        const int n0 = u * kFwdBlockN;
        if constexpr (decltype(masked)::value) {
            const bool need_mask = (n0 + kFwdBlockN > sk) || (CAUSAL && (n0 + kFwdBlockN - 1 > wave_q_lo + delta));
            if (need_mask) {
                const int lim = CAUSAL ? min(sk - 1, m0 + q_row + delta) : sk - 1;
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = n0 + 32 * bi + c_row(r, hi);
                        sacc[bi][r] = key <= lim ? sacc[bi][r] : -INFINITY;
                    }
            }
        }
        // (masked tiles too, except tile 0, which meets an empty running max: its pass would always be thrown away)
        if (kOptimistic && (!decltype(masked)::value || u > 0)) {
            // Optimistic pass: exponentials against the running max as it stands, no row-max reduction.  A lane whose 32 terms sum to
            // <= 2^kPpDeferLog2 holds no term above that bound (the same bound the deferred-max rule below guarantees), so P, l and O
            // stay in range and the pass stands; otherwise (Inf and NaN included) the wave falls through to the exact path, which
            // still has the scores.
            const float mc0 = m_run * c;
            float ps = 0.f;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = fast_exp2(__builtin_fmaf(sacc[bi][r], c, -mc0)), p1 = fast_exp2(__builtin_fmaf(sacc[bi][r + 1], c, -mc0));
                    ps += p0;
                    ps += p1;
                    pf[2 * bi + (r >> 3)][(r & 7) >> 1] = LP<T>::pack2(p0, p1);      // = pack_c_half word by word
                }
            if (__builtin_amdgcn_ballot_w64(!(ps <= 64.0f)) == 0) {
                l_run += ps;
                return;
            }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[bi][r]);
        mx = max_both_halves(mx);
        float psum = 0.f;
        {
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
            // instructions -- measured 5 % SLOWER next to the partner wave's MFMAs)
            const float mc = m_run * c;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(__builtin_fmaf(sacc[bi][r], c, -mc));
                    psum += pv;
                    sacc[bi][r] = pv;
                }
        }
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts) pf[ts] = pack_c_half<T>(sacc[ts >> 1], ts & 1);
        l_run += psum;
    };
    auto advance_ring = [&]() __attribute__((always_inline)) {
        ring_um1 = ring_u;
        ring_u = ring_up1;
        ring_up1 = ring_up1 == 2 ? 0 : ring_up1 + 1;
    };
    // every S phase ends with: this wave's LDS-DMA pieces have landed (vmcnt) -> workgroup barrier
    auto end_s_phase = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // Epilogue, wave-local: normalise, round, stage this wave's 32 rows in LDS (own region, not the rings: the other group may
    // still be reading those), store them as whole rows.  No workgroup barrier: a wave reads back only what it wrote itself.
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        const int rows_here = rows_of(t);
        const int ln = lane;
        const float l_tot = sum_both_halves(l_run);
        // dead rows (row sum exactly 0): O = 0, LSE = 0; a NaN row sum is NOT dead, it propagates (flash_fwd_kernel.h:718,767: `!= 0`)
        const float inv = l_tot != 0.f ? fast_rcp(l_tot) : 0.f;
        const float lse = l_tot != 0.f ? (m_run * c + fast_log2(l_tot)) * kLn2 : 0.f;
        if (hi == 0 && q_row < rows_here) lse_bh[t * kFwdBlockM + q_row] = lse;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w.x = LP<T>::pack2(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
                w.y = LP<T>::pack2(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
                lds_write8(stage, lds_tile_off<D>(q_row, 4 * db + g4) + 8 * hi, w);
            }
        const rsrc_t o_rs = make_rsrc(uniform_ptr(o_ptr_of(t)), (uint32_t)(rows_here - 1) * o_rowb + ROWB);
        constexpr int O_CHUNKS = (32 * SLOTS) / 64;
#pragma unroll
        for (int i = 0; i < O_CHUNKS; ++i) {
            const int chunk = ln + i * 64, row = wave * 32 + chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, lds_read16(stage, lds_tile_off<D>(row, slot)));   // rows >= rows_here fall outside the SRD
        }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    bool prev_active = false;                         // does this wave hold a P tile whose P*V is pending?
    // One iteration = matrix phase M(u) | barrier | softmax phase S(u) | vmcnt(0), barrier.
    auto iteration = [&](int u, auto masked) __attribute__((always_inline)) {
        bool active = true;
        if constexpr (decltype(masked)::value) active = !CAUSAL || (u * kFwdBlockN <= wave_q_hi + delta);
        if (prev_active) pv_step();
        if (active) qk_step();
        issue_dma_k(u);
        __syncthreads();
        issue_dma_v(u);
        if (active) softmax_step(u, masked);
        prev_active = active;
        end_s_phase();
        advance_ring();
    };

    int u = 0;
    if (n_main > 0) {
        iteration(0, no{});
        if (n_main > 1) m_prefetch(ring_um1, ring_u);
        // ring slot of tile u is u % 3: three steps per trip make every slot a constant
        auto step_c = [&](int uu, auto um1, auto u0, auto up1) __attribute__((always_inline)) {
            constexpr int S_UM1 = decltype(um1)::value, S_U = decltype(u0)::value, S_UP1 = decltype(up1)::value;
            {
                u32x4 fr[NST];
                static_for<0, PF>([&](auto jc) { fr[decltype(jc)::value] = pre[decltype(jc)::value]; });
                static_for<0, NST>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (j + PF < NST) fr[j + PF] = m_frag_c(j + PF, S_UM1, S_U);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (FA_STEP_IS_PV(j)) {
                        constexpr int pj = FA_STEP_IDX(j);
                        oacc[FA_PV_DB(pj)] = LP<T>::mfma(fr[j], pf[FA_PV_TS(pj)], oacc[FA_PV_DB(pj)]);
                    } else {
                        constexpr int i = FA_STEP_IDX(j), ks = i / NB, bi = i % NB;
                        if constexpr (ks == 0) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) sacc[bi][r] = 0.f;
                        }
                        sacc[bi] = LP<T>::mfma(fr[j], qf[ks], sacc[bi]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            __syncthreads();
            if (uu + 2 < n_tiles) dma_k_tile(k_srd, uu + 2, S_UM1);
            if (uu + 1 < n_tiles) dma_v_tile(v_srd, uu + 1, S_UP1);
            softmax_step(uu, no{});
#pragma unroll
            for (int j = 0; j < PF; ++j) pre[j] = m_frag_c(j, S_U, S_UP1);
            end_s_phase();
        };
        using i0 = std::integral_constant<int, 0>;
        using i1 = std::integral_constant<int, 1>;
        using i2 = std::integral_constant<int, 2>;
        u = 1;
        if constexpr (D == 128 || BN == 128) {    // (D = 64 with 64-key tiles runs two workgroups per CU on 128 registers: no room for the extra bases)
            for (; u + 3 <= n_main; u += 3) {
                step_c(u, i0{}, i1{}, i2{});
                step_c(u + 1, i1{}, i2{}, i0{});
                step_c(u + 2, i2{}, i0{}, i1{});
            }
        }
        for (; u < n_main; ++u) {                 // the last one or two steady-state tiles (all of them for D = 64)
            m_phase();
            __syncthreads();
            issue_dma_k(u);
            issue_dma_v(u);
            softmax_step(u, no{});
            m_prefetch(ring_u, ring_up1);         // first fragments of M(u+1): V(u) and K(u+1) landed one barrier ago
            end_s_phase();
            advance_ring();
        }
    }
    for (; u < n_tiles; ++u) iteration(u, yes{});        // diagonal / ragged tiles
    // ---- drain: the last P*V and the epilogue ----
    if (prev_active) pv_step();
    epilogue(tile);
    if (group == 0) __syncthreads();          // group A waits for B's last phase (equal barrier counts)
}


template <typename T, int D>
static hipError_t launch_pp_t(const FwdKernelParams& kp, uint32_t grid, hipStream_t stream) {
    if (grid == 0) return hipSuccess;
    if constexpr (D == 64) {
        const bool wide = FA_FWD_D64_BN != 0 ? FA_FWD_D64_BN == 128 : (kp.is_causal && kp.seqlen_k >= kFwdD64WideMinKeys);
        if (wide) {
            if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_pp_kernel<T, 64, true, 128>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
            else hipLaunchKernelGGL((fa_fwd_pp_kernel<T, 64, false, 128>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
            return hipGetLastError();
        }
    }
    if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, true, 64>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
    else hipLaunchKernelGGL((fa_fwd_pp_kernel<T, D, false, 64>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
    return hipGetLastError();
}

// D = 128 has a second forward kernel, fa_fwd_pp16.hip: the same two-group schedule re-tiled for v_mfma_f32_16x16x32 (64 MFMAs per tile
// and wave instead of 32, the same LDS traffic).  The chip runs these kernels against its power cap; the 16x16x32 shape draws less per
// FLOP (tools/powerbench: 1.98 vs 1.66 PFLOP/s on N(0,1) data) and the kernel runs at ~1.87 GHz instead of ~1.55, but it needs ~12 % more
// cycles (twice the MFMA issue slots on the VALU port).  It wins 3-5 % where the cap binds - long launches - and loses 2-10 % on short ones
// (profiles/r3_fwd_mfma16_ab.log), so the launcher picks by the (query, key) pairs PER HEAD, seqlen_q * seqlen_k - not by batch or head count:
// a (batch, head) shard of a problem must get the kernel, hence the bits, the whole problem gets (flash_attn_turing/sharding.py).
// FA_FWD_MFMA16: the policy a process starts with, 0 = never, 1 = always, 2 = by size (default); fa_set_kernel_policy() changes it
// (tests run the whole forward grid through either kernel; a deployment that knows its launches are short can pin the 32x32x16 one).
#ifndef FA_FWD_MFMA16
#define FA_FWD_MFMA16 2
#endif
static std::atomic<int> g_fwd_policy{FA_FWD_MFMA16};
int kernel_policy() { return g_fwd_policy.load(std::memory_order_relaxed); }
int set_kernel_policy(int policy) {
    if (policy < 0 || policy > 2) return -1;
    return g_fwd_policy.exchange(policy, std::memory_order_relaxed);
}
static std::atomic<int64_t> g_policy_bh{0};
int64_t set_policy_problem_heads(int64_t bh) {
    if (bh < 0) return -1;
    return g_policy_bh.exchange(bh, std::memory_order_relaxed);
}
int64_t policy_bh(int64_t b, int64_t h) {
    const int64_t hint = g_policy_bh.load(std::memory_order_relaxed);
    return hint > 0 ? hint : b * h;
}
// Round 6: which head_dim-128 forward is ahead depends on whether the LAUNCH fills the chip, not only on the problem per head: below one workgroup per compute unit the
// clock is not held down by the power cap and the 32x32x16 kernel (half the MFMA instructions) is 5-10 % ahead at every length; from one per unit (two under a causal
// mask, whose workgroups carry half the work on average) the 16x16x32 kernel is 2-9 % ahead from 1k x 1k.  Ratio 16 / 32, fp16, profiles/r6_policy_small_grids.log next to
// profiles/r5_policy_sweep_after_trim.log (workgroups of 256 query rows):
//   no mask:  b1 h8   1k 1.08 (32 wg)  2k 1.07 (64)   4k 1.02 (128)  8k 0.91 (256) |  b1 h32  1k 1.05 (128)  2k 0.94 (256)  4k 0.92 |  b4 h32  512 1.00 (256)  1k 0.965 (512)  2k 0.94
//   causal:   b1 h8   2k 1.10 (64)     4k 1.07 (128)  8k 0.97 (256)                |  b1 h32  2k 1.01 (256)  4k 0.95 (512)  8k 0.93 |  b4 h32  512 1.01 (256)  1k 0.985 (512)  4k 0.96
// Rule: seqlen_q * seqlen_k >= 2^20 AND workgroups >= CUs (causal: >= 2 x CUs, or >= CUs with seqlen_q * seqlen_k >= 2^26).  Rounds 3-5 chose by the pairs per head alone
// (2^20, 2^23 causal) so that a (batch, head) shard got the whole problem's kernel; a shard now states the whole problem's batch x heads (set_policy_problem_heads above).
constexpr int64_t kFwdMfma16MinPairs = (int64_t)1 << 20;             // 1024 x 1024
constexpr int64_t kFwdMfma16LongPairsCausal = (int64_t)1 << 26;      // 8192 x 8192: one workgroup per compute unit is enough under a mask
hipError_t launch_fwd_pp16(const FwdKernelParams& kp, int dtype, uint32_t grid, hipStream_t stream);      // fa_fwd_pp16.hip
int fwd_pp16_block_m(int d);                                                                          // query rows per workgroup of that kernel

// head_dim 64 (round 4): fp16 through the 16x16x32 kernel with 128-key tiles once its row sums ride the matrix pipe - at head_dim 64 the softmax
// VALU work per MFMA is twice that of head_dim 128 and the pipe is half idle, so the adds it takes over are worth more: -2 % at 4k, -5 % at 8k / 16k
// non-causal, -3.6 % at 8k and -5.3 % at 16k causal; +9..24 % at causal 2k-4k, +2 % at non-causal 2k, and bf16 (VALU row sums) +1..2 % everywhere
// (profiles/r4_fwd_d64_mfma16_ab.log; round 3 without the MFMA row sums: -1.4 % at best, profiles/r3_fwd_mfma16_d64_probe.log).
constexpr int64_t kFwdD64Mfma16MinPairs = (int64_t)1 << 24;          // 4096 x 4096
constexpr int64_t kFwdD64Mfma16MinPairsCausal = (int64_t)1 << 26;    // 8192 x 8192
static bool use_mfma16(const FwdKernelParams& kp, int dtype) {
    const int policy = g_fwd_policy.load(std::memory_order_relaxed);
    if (policy == 0 || (kp.d != 128 && kp.d != 64)) return false;
    if (policy == 1) return true;
    if (kp.d == 64) {
        // Round 6: at head_dim 64 the launch matters the other way round.  The 32x32x16 kernel lives off two co-resident workgroups per compute unit; a launch that cannot fill
        // both slots leaves it behind the one-workgroup 16x16x32 kernel by 8-21 % from 2k x 2k (ratio 16 / 32, fp16: no mask b1 h32 2k 0.92, b1 h8 2k 0.975 / 4k 0.92; causal
        // b1 h32 1k 0.965 / 2k 0.91 / 4k 0.79, b1 h8 1k 0.96 .. 4k 0.86), while full launches keep the round-4 thresholds (b4 h32: 1.02-1.11 at 2k-4k causal, b16 h32: 1.07-1.33
        // below 4k): profiles/r6_policy_d64_before.log / r6_policy_d64.log.  bf16 keeps its VALU row sums and stays on the 32x32x16 kernel.
        if (dtype != 0) return false;
        const int64_t pairs = (int64_t)kp.seqlen_q * kp.seqlen_k, wgs = policy_bh(kp.b, kp.h) * (((int64_t)kp.seqlen_q + 255) / 256), cus = device_cu_count();
        if (kp.is_causal) return pairs >= kFwdD64Mfma16MinPairsCausal || (pairs >= ((int64_t)1 << 22) && wgs <= 2 * cus) || (pairs >= ((int64_t)1 << 20) && 2 * wgs <= cus);
        return pairs >= kFwdD64Mfma16MinPairs || (pairs >= ((int64_t)1 << 22) && wgs <= cus);
    }
    // (packed sequences: max_seqlen_q x max_seqlen_k and the plain grid's workgroup count; never total_q - the optional hint must not change which kernel,
    // hence which bits, a call gets: tests/test_fuzz_gpu.py compares the compact and the plain varlen grid bit for bit)
    const int64_t pairs = (int64_t)kp.seqlen_q * kp.seqlen_k;
    const int64_t wgs = policy_bh(kp.b, kp.h) * (((int64_t)kp.seqlen_q + 255) / 256), cus = device_cu_count();
    if (pairs < kFwdMfma16MinPairs) return false;
    return kp.is_causal ? (wgs >= 2 * cus || (wgs >= cus && pairs >= kFwdMfma16LongPairsCausal)) : wgs >= cus;
}

// the kernel that serves the LARGE problems of a head dimension (what a profile of the BASELINE configurations shows)
// (head_dim 64: of fp16 inputs; bf16 stays on fa_fwd_pp_kernel unless the 16x16x32 set is pinned - fwd_kernel_name_for answers per dtype)
const char* fwd_kernel_name(int d) { return (d == 128 || d == 64) && g_fwd_policy.load(std::memory_order_relaxed) != 0 ? "fa_fwd_pp16_kernel" : "fa_fwd_pp_kernel"; }

const char* fwd_kernel_name_for(const FwdKernelParams& kp, int dtype) { return use_mfma16(kp, dtype) ? "fa_fwd_pp16_kernel" : "fa_fwd_pp_kernel"; }

// (Round 4 built a third head_dim-128 forward, one wave per SIMD with 64 query rows per wave and O / Q in asm-owned accumulation registers,
// fa_fwd_w4.hip: bit-identical to fa_fwd_pp16 and 7-12 % slower - a lone wave cannot issue 16x16x32 MFMAs at the pipe's rate.  Not in the
// product; profiles/r4_fwd_w4_one_wave_per_simd_ab.log, file in the history.)
hipError_t launch_fwd(FwdKernelParams kp, int dtype, hipStream_t stream) {
    const int block_m = use_mfma16(kp, dtype) ? fwd_pp16_block_m(kp.d) : kFwdBlockM;
    kp.n_q_tiles = (uint32_t)((kp.seqlen_q + block_m - 1) / block_m);
    kp.varlen_slots = kp.cu_seqlens_q != nullptr ? varlen_slot_count(kp.total_q, kp.b, block_m, kp.n_q_tiles) : 0u;
    const uint32_t grid = kp.varlen_slots != 0 ? kp.varlen_slots * (uint32_t)kp.h : kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    // (head_dim 64 runs two workgroups per compute unit except in its 128-key causal shape, launch_pp_t)
    const int wg_per_cu = (kp.d == 64 && !use_mfma16(kp, dtype) && !(FA_FWD_D64_BN != 0 ? FA_FWD_D64_BN == 128 : (kp.is_causal && kp.seqlen_k >= kFwdD64WideMinKeys))) ? 2 : 1;
    kp.group_heads = causal_group_heads(kp.is_causal != 0, kp.varlen_slots != 0 ? kp.b : 0, kp.varlen_slots != 0 ? kp.h : (int64_t)kp.b * kp.h, kp.seqlen_q, kp.seqlen_k, kp.n_q_tiles, wg_per_cu, (int64_t)4 * kp.seqlen_k * kp.d);
    if (use_mfma16(kp, dtype)) return launch_fwd_pp16(kp, dtype, grid, stream);
    if (dtype == 0) return kp.d == 128 ? launch_pp_t<_Float16, 128>(kp, grid, stream) : launch_pp_t<_Float16, 64>(kp, grid, stream);
    return kp.d == 128 ? launch_pp_t<__bf16, 128>(kp, grid, stream) : launch_pp_t<__bf16, 64>(kp, grid, stream);
}

}  // namespace fa
