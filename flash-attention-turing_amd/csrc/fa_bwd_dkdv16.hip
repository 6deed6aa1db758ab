// fa_bwd_dkdv16.hip -- the dK/dV kernel of fa_bwd.hip (same workgroup shape, resident K / V tiles, Q / dO rings by hand-issued LDS-DMA,
// per-head stream descriptors, statistics ring, AGPR accumulators, head-group split; read its header first) re-tiled for
// v_mfma_f32_16x16x32_{f16,bf16}, head_dim 128 only.  Why: fa_bwd_dq16.hip / fa_fwd_pp16.hip (the power cap and the MFMA shape).
//
// Layout: wave w owns key block kb = w & 3 (32 keys) and the 32-row half qh = w >> 2 of every Q / dO tile, as before.  A lane is
// (k-group g = lane >> 4, column n = lane & 15) and owns TWO keys of its wave's 32: key column kc (0 / 1) is key 16*kc + kPerm[n]
// (kPerm = the row permutation that keeps the swizzled tile image conflict-free under 16-row fragment reads).  S = Q K^T and
// dP = dO V^T come in 16 x 16 blocks [query block i of the half][key column kc], contraction in 4 steps of 32 d, A = Q / dO rows from
// LDS (one fragment feeds the two key columns), B = K / V fragments (K: all in registers, V: the first step).  Block (i, kc) holds query
// rows 4*kPi2[g] + r of block i: a lane's P / dS values of blocks i = 0 and 1 ARE the 8 k-slots of the B operand of
// dV^T += dO^T P and dK^T += Q^T dS over the half's 32 query rows (A = transposed LDS reads, one fragment feeds both key columns).
#include <type_traits>

#include "fa_bwd_dkdv_common.hpp"

namespace fa {

#ifndef FA_KV16_VREG
#define FA_KV16_VREG(CAUSAL) ((CAUSAL) ? 1 : 0)      // k-steps (of 32 d) of V held in registers next to all 4 of K: whatever leaves the instance's tile
#endif                                               // loop without a spill (a scratch reload per tile cost the non-causal instance 7 %)
#ifndef FA_KV16_TOGGLE
#define FA_KV16_TOGGLE 1      // ring-slot addressing: per-lane base registers toggled once per tile (13 VALU) instead of base + slot offset per read (34)
#endif
#ifndef FA_KV16_RV
#define FA_KV16_RV 0          // (with the toggle) V row reads through their own absolute base registers: 4 registers for 6 address adds
#endif
#ifndef FA_KV16_PF
#define FA_KV16_PF 4          // transposed fragments in flight in the dV / dK phase (3: one spill op per tile in the causal instances; 2-4 time the same)
#endif

#ifndef FA_KV16_DMA_EARLY
#define FA_KV16_DMA_EARLY 0   // which q-half group requests the next tile at the top of the loop (the other one after its S / dP MFMAs); 2 = both at the top
#endif
// (Round 6, measured and not kept - the switches were never part of a committed product build; the logs say what was built:
//   FA_KV16_DMA_STAGGER, three / four request points per tile instead of two: -2.3..+2.5 % (three) / +11..19 % (four), profiles/r6_dkdv_dma_stagger_ab.log;
//   FA_KV16_SKEW, the two q-half groups a whole phase apart (one instruction stream, waves 4-7 with their barrier behind the score phase, waves 0-3 behind the dV / dK phase, three-slot
//   rings): bit-identical, -0.4..+8.7 %, profiles/r6_dkdv_skew_ab.log - the pair is not in lock-step to begin with: at priority 1 waves 4-7 run ~800 cycles ahead inside a tile and
//   wait at the barrier (profiles/r6_dkdv16_phase_timing.log), and hipcc moves the exponentials into the dV / dK phase's fragment prologue;
//   FA_KV16_Q1_DUTIES, all requests (and the statistics) by waves 4-7, which have that slack: +0.2..+12 %, profiles/r6_dkdv_q1_duties_ab.log.)
// (Round 5, measured and not kept - code in the history at 1ef1631: FA_KV16_DMA_SPREAD, the four pieces issued one by one between the MFMAs of the S / dP k-steps
// (+1.5..6 %) or of the dV / dK fragment steps (+5..23 %) instead of back to back, profiles/r5_bwd_dkdv_dma_spread_ab.log.)
#ifndef FA_KV16_STAT_PRED
#define FA_KV16_STAT_PRED(CAUSAL) (!(CAUSAL))   // the per-tile statistics load under an EXEC mask in the six waves that do not use it: the non-causal instances only
#endif
#ifndef FA_KV16_ABL
#define FA_KV16_ABL 0         // timing-only ablations (results are WRONG), bit mask: 1 no workgroup barrier at the end of a tile, 2 no row-fragment LDS reads in the
#endif                        // S / dP phase, 8 no exponentials, 16 no LDS-DMA of the next tile, 32 nothing of the next tile is waited for (no vmcnt wait in the loop at all; the statistics are not refreshed) (profiles/r4_bwd_dkdv16_ablations.log)
#ifndef FA_KV16_DMA_DEBUG
#define FA_KV16_DMA_DEBUG 0   // test builds only (tests/test_dma_protocol_gpu.py; the product is 0 and its ISA does not change): 1 = LATE ISSUE, the next tile's LDS-DMA goes
#endif                        // out directly in front of the wait that retires it (the bytes land as late as the ring protocol can tolerate); 2 = waves 4-7 sleep ~4000 cycles first
#ifndef FA_KV16_NOP_ONCE
#define FA_KV16_NOP_ONCE 1    // the VALU -> MFMA source hazard of the asm-issued dV / dK MFMAs (P / dS come straight from v_cvt_pk) is padded once, in
#endif                        // front of the phase, instead of with an s_nop in front of each of its 32 MFMAs: -0.2..-0.8 %
#if FA_KV16_NOP_ONCE
#define FA_KV16_NOP ""
#else
#define FA_KV16_NOP "s_nop 1\n\t"
#endif
template <typename T>
struct LP16;
template <>
struct LP16<_Float16> {
    // accumulate into the accumulator half of the register file (see LP<T>::mfma_agpr); s_nop 1: a / b may have just been written by VALU
    static FA_DEV void mfma_agpr(f32x4& acc, u32x4 a, u32x4 b) { asm(FA_KV16_NOP "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
};
template <>
struct LP16<__bf16> {
    static FA_DEV void mfma_agpr(f32x4& acc, u32x4 a, u32x4 b) { asm(FA_KV16_NOP "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
};

// FA_KV16_MIN_WAVES: waves per SIMD the register budget is cut for (head_dim 64, round 5: 2 = one workgroup per compute unit, accumulators in
// 64 AGPRs + up to 192 VGPRs; 4 = two co-resident workgroups on 128 registers in all, what the 32x32x16 head_dim-64 kernel runs with)
#ifndef FA_KV16_MIN_WAVES
#define FA_KV16_MIN_WAVES(D) 2
#endif
#ifndef FA_KV16_D64_PF
#define FA_KV16_D64_PF 2      // head_dim 64: transposed fragments in flight in the dV / dK phase (8 steps there; 2 against 4: -0.5..-2.5 %, profiles/r5_bwd_d64_mfma16_ab.log)
#endif
#ifndef FA_KV16_D64_VREG
#define FA_KV16_D64_VREG 2    // head_dim 64: both k-steps of V live in registers next to both of K (16 + 16 registers)
#endif

// -DFA_KV_TIMING (development aid, tools/phase_timing_dkdv.py --layout 16): per-wave s_memtime stamps at the phase edges of the tile loop, summed per phase, plus the
// workgroup's prologue / epilogue cycles and its start / end on the 100 MHz wall clock, left in p.ws[(block * 8 + wave) * 16 + ...] (the caller passes a workspace)
#ifdef FA_KV_TIMING
#define FA_KV16_STAMP(i) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define FA_KV16_STAMP(i) do { } while (0)
#endif
template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(kKvThreads, FA_KV16_MIN_WAVES(D)) void fa_bwd_dkdv16_kernel(const BwdKernelParams p) {
#ifdef FA_KV_TIMING
    const uint64_t t_start = __builtin_amdgcn_s_memtime(), rt_start = __builtin_amdgcn_s_memrealtime();
#endif
    static_assert(D == 128 || D == 64, "head_dim");
    constexpr int KS = D / 32, DB = D / 16, ROWB = D * 2, SLOTS = D / 8;
    constexpr int KVB = kKvBlockN * ROWB;                   // the workgroup's K (or V) tile
    constexpr int TILEB = kKvBlockM * ROWB;                 // one Q (or dO) tile
    constexpr int STATB = 2 * kKvBlockM * 4;                // lse2 + dsum of one tile
    constexpr int OFF_V = KVB, OFF_Q = 2 * KVB, OFF_DO = 2 * KVB + 2 * TILEB, OFF_STAT = 2 * KVB + 4 * TILEB;
    // aligned to the toggle span: the ring-slot toggle XORs ABSOLUTE LDS addresses with TILEB / STATB, which is only a slot switch when the
    // array starts at a multiple of 2 * TILEB (it is the kernel's only __shared__ object and sat at 0 anyway)
    __shared__ __attribute__((aligned(32768))) char smem_raw[OFF_STAT + 2 * STATB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* ktile = smem;
    FA_LDS char* vtile = smem + OFF_V;
    FA_LDS char* stat = smem + OFF_STAT;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave & 3, qh = wave >> 2;             // key block / q-half of this wave
    const int pi_g = (0x2130 >> (4 * g)) & 3;             // d chunks  {0, 3, 1, 2}
    const int pi2_g = (0x3120 >> (4 * g)) & 3;            // row sub-blocks {0, 2, 1, 3}
    const int perm_n = 4 * ((0x3120 >> (4 * (n16 >> 2))) & 3) + (n16 & 3);      // kPerm[n16]: row of a 16-row block that fragment row / column n16 stands for

    int tile, batch, vhead, tiles_seq;
    if (!decode_work<kKvBlockN>(blockIdx.x, p.n_k_tiles, p.varlen_slots, p.cu_seqlens_k, p.b, p.h_k * p.n_split, tile, batch, vhead, tiles_seq, p.group_heads)) return;
    const int head_k = vhead / p.n_split, split = vhead - head_k * p.n_split;
    const int heads_here = p.h_ratio / p.n_split;            // query heads of this workgroup

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch, v_boff = (int64_t)batch * p.v.batch,
            do_boff = (int64_t)batch * p.dout.batch, dk_boff = (int64_t)batch * p.dk.batch, dv_boff = (int64_t)batch * p.dv.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int qb = p.cu_seqlens_q[batch], kb_ = p.cu_seqlens_k[batch];
        sq = min(p.cu_seqlens_q[batch + 1] - qb, p.seqlen_q);   // clamp to the declared max_seqlen_q (padded LSE / D rows)
        sk = p.cu_seqlens_k[batch + 1] - kb_;
        q_row0 = qb; k_row0 = kb_;
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

    const int key_loc = kb * 32 + perm_n;                // this lane's first key inside the 128-key block (the second: + 16)
    const int wave_k_lo = n0 + kb * 32, wave_k_hi = wave_k_lo + 31;

    // ---- LDS-DMA tables (fa_bwd.hip) ----------------------------------------------------------------------------------------------
    constexpr int PPW_KV = (kKvBlockN * SLOTS / 64) / 8;
    constexpr int PPW_Q = (kKvBlockM * SLOTS / 64) / 8;
    auto piece_src = [&](int piece, uint32_t rowb) {
        const int chunk = piece * 64 + lane, row = chunk / SLOTS, phys = chunk % SLOTS;
        return (uint32_t)row * rowb + lds_tile_logical_slot<D>(row, phys) * 16;
    };
    const uint32_t lds0 = lds_addr(smem);
    uint32_t q_src[PPW_Q], do_src[PPW_Q];
#pragma unroll
    for (int i = 0; i < PPW_Q; ++i) {
        q_src[i] = piece_src(wave * PPW_Q + i, q_rowb);
        do_src[i] = piece_src(wave * PPW_Q + i, do_rowb);
    }

    // row reads (A of S / dP: Q / dO rows; B of dP: V rows): fragment row / column n16 -> row kPerm[n16] of a 16-row block, slot 4*ks + kPi[g]
    uint32_t row_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) row_rd[ks] = lds_tile_off<D>(perm_n, 4 * ks + pi_g);
    // transposed reads (A of dV^T / dK^T: dO^T / Q^T): lane group g points at the 4 rows 4*kPi2[g] .. +3 of a 16-row block, 16 d wide
    uint32_t tr_rd[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) tr_rd[db] = lds_tile_off<D>(4 * pi2_g + (n16 >> 2), 2 * db + ((n16 & 3) >> 1)) + 8 * (n16 & 1);
    const float c = p.scale_log2e;
#if FA_KV16_TOGGLE
    // Q / dO ring reads as absolute LDS addresses of ring slot 0 with everything wave- or lane-dependent folded in; a read is base register +
    // immediate (dO = Q + 2 * TILEB), and the registers move to the other slot by flipping ONE address bit per tile: the slots are TILEB =
    // 2^14 apart and every offset inside a tile stays below that.  Same for the statistics ring (2^9 apart).
    static_assert((TILEB & (TILEB - 1)) == 0 && OFF_Q % (2 * TILEB) == 0 && OFF_DO == OFF_Q + 2 * TILEB && STATB == (1 << 9) && OFF_STAT % (2 * STATB) == 0, "slot toggle");
    uint32_t rq[KS], rt[DB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rq[ks] = lds0 + OFF_Q + row_rd[ks] + (uint32_t)(32 * qh) * ROWB;
#pragma unroll
    for (int db = 0; db < DB; ++db) rt[db] = lds0 + OFF_Q + tr_rd[db] + (uint32_t)(32 * qh) * ROWB;
    uint32_t rs = lds0 + OFF_STAT + (uint32_t)(32 * qh + 4 * pi2_g) * 4;
#if FA_KV16_RV
    uint32_t rv[KS];                                         // V rows of this wave's key block (no ring: constant)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rv[ks] = lds0 + OFF_V + row_rd[ks] + (uint32_t)(kb * 32) * ROWB;
#endif
    auto at = [](uint32_t a) __attribute__((always_inline)) { return (const FA_LDS char*)(uintptr_t)a; };
#endif

    f32x4 dkacc[DB][2], dvacc[DB][2];                     // dK^T / dV^T: d rows 16*db + 4*g + r, key column kc
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) { dkacc[db][kc] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[db][kc] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int head_first = head_k * p.h_ratio + split * heads_here;
    srd_t q_srd = make_srd(nullptr, 0), do_srd = make_srd(nullptr, 0);
    rsrc_t st_rs = make_rsrc(nullptr, 0);
    auto set_head = [&](int hq) {
        const T* qb = uniform_ptr((const T*)p.q_ptr + q_boff + q_row0 * p.q.row + (int64_t)hq * p.q.head);
        const T* dob = uniform_ptr((const T*)p.do_ptr + do_boff + q_row0 * p.dout.row + (int64_t)hq * p.dout.head);
        q_srd = make_srd(qb, sq > 0 ? (uint32_t)(sq - 1) * q_rowb + ROWB : 0u);          // rows past the end of the sequence read zeros
        do_srd = make_srd(dob, sq > 0 ? (uint32_t)(sq - 1) * do_rowb + ROWB : 0u);
        const float* sb = uniform_ptr((wave == 0 ? p.lse_ptr : p.dsum_ptr) + ((int64_t)batch * p.h + hq) * p.lse_row_stride);
        st_rs = make_rsrc(sb, wave < 2 ? (uint32_t)sq * 4u : 0u);
    };
    int pf_head = head_first, pf_tile = 0;
    auto pf_m0 = [&]() { return (qt_begin + pf_tile) * kKvBlockM; };
    auto pf_advance = [&]() {
        if (++pf_tile == tiles_per_head) { pf_tile = 0; ++pf_head; if (pf_head < head_first + heads_here) set_head(pf_head); }
    };
    auto issue_tile = [&](int buf) {
        const uint32_t m0 = (uint32_t)pf_m0();
#pragma unroll
        for (int i = 0; i < PPW_Q; ++i) {
            const int piece = wave * PPW_Q + i;
            dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(q_srd, q_src[i] + m0 * q_rowb, lds0 + OFF_Q + buf * TILEB + piece * 1024);
            dma16_to_lds_hidden<FA_BWD_DMA_SAVE_M0 != 0>(do_srd, do_src[i] + m0 * do_rowb, lds0 + OFF_DO + buf * TILEB + piece * 1024);
        }
    };
    const float st_mult = wave == 0 ? -kLog2e : -1.0f;
    auto load_stat = [&](bool valid) -> float {
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

    // B operands that never change: this wave's K fragments (all 4 k-steps, both key columns) and the first VREG k-steps of V
    constexpr int VREG_WANT = D == 64 ? FA_KV16_D64_VREG : FA_KV16_VREG(CAUSAL);
    constexpr int VREG = VREG_WANT < KS ? VREG_WANT : KS;
    u32x4 kreg[KS][2], vreg[VREG > 0 ? VREG : 1][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) kreg[ks][kc] = lds_read16(ktile, row_rd[ks] + (kb * 32 + 16 * kc) * ROWB);
#pragma unroll
    for (int ks = 0; ks < VREG; ++ks)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) vreg[ks][kc] = lds_read16(vtile, row_rd[ks] + (kb * 32 + 16 * kc) * ROWB);

    // FA_KV_PRIO (round 4): issue priority between the two q-half groups, which share every SIMD pairwise.  Waves 0-3 are the older ones,
    // win every arbitration and wait ~1000 of ~3700 cycles per tile at the barrier (profiles/NOTEBOOK.md 3c).  1 (shipped) = waves 4-7 at priority 1 for
    // the whole loop: -1.1..-3.4 % on every shape but causal 2k (+0.3 %); 2 = the groups alternate tile by tile: +0..2 %; 3 = waves 0-3 at
    // priority 1 (control): -1..+1 %.  profiles/r4_bwd_dkdv_prio_ab.log.  (+1 inside the MFMA clusters, with or without the static offset: +1..6 %,
    // profiles/r4_prio_mfma_phase_ab.log.)
#ifndef FA_KV_PRIO
#define FA_KV_PRIO 1
#endif
#if FA_KV_PRIO == 1
    if (qh == 1) __builtin_amdgcn_s_setprio(1);
#elif FA_KV_PRIO == 3
    if (qh == 0) __builtin_amdgcn_s_setprio(1);
#endif
    int cur_tile = 0;
#ifdef FA_KV_TIMING
    uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const uint64_t t_loop = tlast;
#endif
    for (int it = 0; it < n_iters; ++it) {
#if FA_KV_PRIO == 2
        if ((it ^ qh) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        const int m0 = (qt_begin + cur_tile) * kKvBlockM;
        if (++cur_tile == tiles_per_head) cur_tile = 0;
        const int buf = it & 1;
#if !FA_KV16_TOGGLE
        FA_LDS char* qbuf = smem + OFF_Q + buf * TILEB;
        FA_LDS char* dobuf = smem + OFF_DO + buf * TILEB;
        FA_LDS char* sbuf = stat + buf * STATB;
#endif
        const bool more = (it + 1 < n_iters);
#if !(FA_KV16_ABL & 16) && FA_KV16_DMA_DEBUG != 1
        if (more && (FA_KV16_DMA_EARLY == 2 || qh == FA_KV16_DMA_EARLY)) issue_tile(buf ^ 1);          // ring slot buf^1 was last read in iteration it-1; waves 4-7 issue after their S / dP MFMAs
#endif
        // Only the two statistics waves need this load, and a vector-memory instruction costs its wave ~100 cycles whatever it returns.  A LANE-dependent condition
        // (an EXEC mask around the instruction) in the non-causal instances: -0.6..-1.2 %; under a mask 0..+1.3 %, so those keep the load in all eight waves; the
        // wave-uniform `wave < 2` (a scalar branch) cost the non-causal instances 8-12 % (profiles/r4_bwd_dkdv16_ablations.log 6, 7)
        float st_next = 0.f;
        if (!FA_KV16_STAT_PRED(CAUSAL) || tid < 128) st_next = load_stat(more);
        FA_KV16_STAMP(0);                                   // top of the tile: LDS-DMA requests (waves 0-3), statistics load

        const int mh = m0 + 32 * qh;                       // first query row of this wave's half
        // (no wave-level causal skip, on purpose: fa_bwd.hip.  A fully masked wave-tile runs the body with P = dS = 0 by select.)
        {
            // causal mask: element (i, kc, r) is query row mh + 16*i + 4*kPi2[g] + r and key n0 + key_loc + 16*kc; visible iff key <= row + delta
            //   <=>  16*i + r >= thr[kc]  with everything tile- / lane-dependent folded once per tile
            const bool need_mask = CAUSAL && (wave_k_hi > mh + delta);
            int thr[2];
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) thr[kc] = (n0 + key_loc + 16 * kc) - (mh + 4 * pi2_g + delta);
            f32x4 nd4[2], nl4[2];                           // -D and -LSE*log2(e) of this lane's query rows of block i
#pragma unroll
            for (int i = 0; i < 2; ++i)
#if FA_KV16_TOGGLE
                nd4[i] = *(const FA_LDS f32x4*)(at(rs) + kKvBlockM * 4 + 16 * i * 4);
#else
                nd4[i] = *(const FA_LDS f32x4*)(sbuf + kKvBlockM * 4 + (32 * qh + 16 * i + 4 * pi2_g) * 4);
#endif
            f32x4 sacc[2][2], dpacc[2][2];                  // [query block i][key column kc]
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 qa[2], da[2], vf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#if FA_KV16_ABL & 2
                    qa[i] = kreg[ks][i]; da[i] = kreg[ks][i ^ 1];
#elif FA_KV16_TOGGLE
                    qa[i] = lds_read16(at(rq[ks]), 16 * i * ROWB);
                    da[i] = lds_read16(at(rq[ks]), 2 * TILEB + 16 * i * ROWB);
#else
                    qa[i] = lds_read16(qbuf, row_rd[ks] + (32 * qh + 16 * i) * ROWB);
                    da[i] = lds_read16(dobuf, row_rd[ks] + (32 * qh + 16 * i) * ROWB);
#endif
                }
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#if FA_KV16_ABL & 2
                    vf[kc] = ks < VREG ? vreg[ks < VREG ? ks : 0][kc] : kreg[ks][kc ^ 1];
#elif FA_KV16_TOGGLE && FA_KV16_RV
                    vf[kc] = ks < VREG ? vreg[ks < VREG ? ks : 0][kc] : lds_read16(at(rv[ks]), 16 * kc * ROWB);
#else
                    vf[kc] = ks < VREG ? vreg[ks < VREG ? ks : 0][kc] : lds_read16(vtile, row_rd[ks] + (kb * 32 + 16 * kc) * ROWB);
#endif
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        sacc[i][kc] = LP<T>::mfma16(qa[i], kreg[ks][kc], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sacc[i][kc]);      // S = Q K^T
                        dpacc[i][kc] = LP<T>::mfma16(da[i], vf[kc], ks == 0 ? nd4[i] : dpacc[i][kc]);                              // dP - D = dO V^T - D
                    }
            }
            FA_KV16_STAMP(1);                               // S / dP MFMAs (with their row-fragment reads)
#if FA_KV16_DMA_DEBUG == 2
            if (more && qh == 1) asm volatile("s_sleep 64" ::: "memory");
#endif
#if !(FA_KV16_ABL & 16) && FA_KV16_DMA_DEBUG != 1
            if (more && FA_KV16_DMA_EARLY != 2 && qh == (FA_KV16_DMA_EARLY ^ 1)) issue_tile(buf ^ 1);      // waves 4-7 request their pieces HERE, while waves 0-3 are still in their S / dP MFMAs
#endif
#if FA_KV16_DMA_DEBUG != 1
            if (more) pf_advance();
#endif
            FA_KV16_STAMP(2);                               // LDS-DMA requests (waves 4-7), descriptors of the next tile
#pragma unroll
            for (int i = 0; i < 2; ++i)
#if FA_KV16_TOGGLE
                nl4[i] = *(const FA_LDS f32x4*)(at(rs) + 16 * i * 4);
#else
                nl4[i] = *(const FA_LDS f32x4*)(sbuf + (32 * qh + 16 * i + 4 * pi2_g) * 4);
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
#if FA_KV16_ABL & 8
                    for (int r = 0; r < 4; ++r) sacc[i][kc][r] = sacc[i][kc][r];
#else
                    for (int r = 0; r < 4; ++r) sacc[i][kc][r] = fast_exp2(__builtin_fmaf(sacc[i][kc][r], c, nl4[i][r]));      // P (flash_bwd_kernel.h:1329)
#endif
            if (need_mask) {                                        // wave-uniform branch: only diagonal tiles pay for the select
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[i][kc][r] = (16 * i + r >= thr[kc]) ? sacc[i][kc][r] : 0.f;
            }
            // P and dS = P * (dP - D) (:1354), rounded (:1359-1360), as the B operands of the half's 32-query contraction
            u32x4 pfr[2], dsfr[2];
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                pfr[kc].x = LP<T>::pack2(sacc[0][kc][0], sacc[0][kc][1]);
                pfr[kc].y = LP<T>::pack2(sacc[0][kc][2], sacc[0][kc][3]);
                pfr[kc].z = LP<T>::pack2(sacc[1][kc][0], sacc[1][kc][1]);
                pfr[kc].w = LP<T>::pack2(sacc[1][kc][2], sacc[1][kc][3]);
                dsfr[kc].x = LP<T>::pack2(sacc[0][kc][0] * dpacc[0][kc][0], sacc[0][kc][1] * dpacc[0][kc][1]);
                dsfr[kc].y = LP<T>::pack2(sacc[0][kc][2] * dpacc[0][kc][2], sacc[0][kc][3] * dpacc[0][kc][3]);
                dsfr[kc].z = LP<T>::pack2(sacc[1][kc][0] * dpacc[1][kc][0], sacc[1][kc][1] * dpacc[1][kc][1]);
                dsfr[kc].w = LP<T>::pack2(sacc[1][kc][2] * dpacc[1][kc][2], sacc[1][kc][3] * dpacc[1][kc][3]);
            }
            // dV^T += dO^T P and dK^T += Q^T dS: 2*DB fragments (transposed LDS reads), each feeding the two key columns.  The accumulators
            // are inline-asm operands, so the reads are software-pipelined by hand - fragment j + PF is requested before the MFMAs of j.
            constexpr int NST = 2 * DB, PF = D == 64 ? FA_KV16_D64_PF : FA_KV16_PF;            // step j = (db, which): which 0 -> dV (dO^T), 1 -> dK (Q^T)
            auto rd_frag = [&](int j) {
                const int db = j >> 1;
#if FA_KV16_TOGGLE
                const uint32_t o = (j & 1) ? 0u : 2u * TILEB;
                const u32x2 a0 = lds_read_tr8(at(rt[db]), o);
                const u32x2 a1 = lds_read_tr8(at(rt[db]), o + 16 * ROWB);
#else
                FA_LDS char* src = (j & 1) ? qbuf : dobuf;
                const u32x2 a0 = lds_read_tr8(src, tr_rd[db] + (32 * qh) * ROWB);
                const u32x2 a1 = lds_read_tr8(src, tr_rd[db] + (32 * qh + 16) * ROWB);
#endif
                return u32x4{a0.x, a0.y, a1.x, a1.y};
            };
            FA_KV16_STAMP(3);                               // exponentials, mask, dS, conversions
            u32x4 frag[NST];
#pragma unroll
            for (int j = 0; j < PF; ++j) frag[j] = rd_frag(j);
#if FA_KV16_NOP_ONCE
            // P / dS were just written by VALU (v_cvt_pk): the MFMAs below are inline asm, the hazard recogniser does not see them read those registers
            asm volatile("s_nop 3" : "+v"(pfr[0]), "+v"(pfr[1]), "+v"(dsfr[0]), "+v"(dsfr[1]));
#endif
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                if (j + PF < NST) frag[j + PF] = rd_frag(j + PF);
                __builtin_amdgcn_sched_barrier(0);
                const int db = j >> 1;
                if (j & 1) { LP16<T>::mfma_agpr(dkacc[db][0], frag[j], dsfr[0]); LP16<T>::mfma_agpr(dkacc[db][1], frag[j], dsfr[1]); }      // dK^T += Q^T dS
                else { LP16<T>::mfma_agpr(dvacc[db][0], frag[j], pfr[0]); LP16<T>::mfma_agpr(dvacc[db][1], frag[j], pfr[1]); }              // dV^T += dO^T P
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        FA_KV16_STAMP(4);                                   // dV / dK MFMAs (with their transposed reads)
#if FA_KV16_DMA_DEBUG == 1
        if (more) { issue_tile(buf ^ 1); pf_advance(); }
#endif
#if !(FA_KV16_ABL & 32)
        if (more && wave < 2) store_stat(st_next, buf ^ 1);
        asm volatile("" :: "v"(st_next));               // consumed on every path: hipcc never has to guard the register at the loop top
#endif
#if !(FA_KV16_ABL & 32)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's DMA pieces (and statistics) have landed
#endif
        FA_KV16_STAMP(5);                                   // statistics store + vmcnt(0): the next tile's pieces have landed
#if FA_KV16_TOGGLE
        // on to the other ring slot (inline asm: hipcc would otherwise re-derive the addresses from `buf` and put the adds back)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(rq[ks]) : "s"((uint32_t)TILEB));
#pragma unroll
        for (int db = 0; db < DB; ++db) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(rt[db]) : "s"((uint32_t)TILEB));
        asm volatile("v_xor_b32 %0, %1, %0" : "+v"(rs) : "s"((uint32_t)STATB));
#endif
#if !(FA_KV16_ABL & 1)
        __syncthreads();
#endif
        FA_KV16_STAMP(6);                                   // ring-slot toggle + workgroup barrier
    }
#ifdef FA_KV_TIMING
    const uint64_t t_loop_end = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: the loop's last barrier has passed, K / V tiles, rings and stats are dead, LDS is scratch (fa_bwd_dkdv_common.hpp
    // describes the steps; the accumulator layout differs: registers [db][kc][r] = d 16*db + 4*g + r of key key_loc + 16*kc) ----------
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) { asm volatile("" : "+a"(dkacc[db][kc])); asm volatile("" : "+a"(dvacc[db][kc])); }
    constexpr int XR = DB * 8;                                                 // accumulator registers per lane and tensor (64)
    FA_LDS float* xch = (FA_LDS float*)smem;                                   // 4 key blocks x XR x 64 lanes floats (64 KiB)
    FA_LDS char* out_t = smem + 4 * XR * 64 * 4;                               // staged output tile (32 KiB)
    static_assert(4 * XR * 64 * 4 + kKvBlockN * ROWB <= OFF_STAT, "epilogue scratch must fit the kernel's LDS");
    constexpr int O_CHUNKS = (kKvBlockN * SLOTS) / kKvThreads;
    const uint32_t dk_rowb = (uint32_t)(p.dk.row * 2), dv_rowb = (uint32_t)(p.dv.row * 2);
    const rsrc_t dk_rs = make_rsrc(dk_base, (uint32_t)(keys_here - 1) * dk_rowb + ROWB);
    const rsrc_t dv_rs = make_rsrc(dv_base, (uint32_t)(keys_here - 1) * dv_rowb + ROWB);
    auto hand_over = [&](f32x4 (&acc)[DB][2]) __attribute__((always_inline)) {      // the qh = 1 waves pass their partial sums to their qh = 0 partner
        if (qh == 1) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[(kb * XR + (db * 2 + kc) * 4 + r) * 64 + lane] = acc[db][kc][r];
        }
        __syncthreads();
    };
    auto reduce_and_store = [&](f32x4 (&acc)[DB][2], float mult, rsrc_t rs, uint32_t rowb) __attribute__((always_inline)) {
        hand_over(acc);
        if (qh == 0) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    float v4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v4[r] = (acc[db][kc][r] + xch[(kb * XR + (db * 2 + kc) * 4 + r) * 64 + lane]) * mult;
                    u32x2 w;
                    w.x = LP<T>::pack2(v4[0], v4[1]);
                    w.y = LP<T>::pack2(v4[2], v4[3]);
                    lds_write8(out_t, lds_tile_off<D>(key_loc + 16 * kc, 2 * db + (g >> 1)) + 8 * (g & 1), w);      // d = 16*db + 4*g + {0..3}
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < O_CHUNKS; ++i) {
            const int chunk = tid + i * kKvThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(rs, (uint32_t)row * rowb + slot * 16, lds_read16(out_t, lds_tile_off<D>(row, slot)));
        }
        __syncthreads();                                                       // scratch is reused by the next tensor
    };
    if (p.n_split == 1) {
        reduce_and_store(dkacc, p.scale, dk_rs, dk_rowb);
        reduce_and_store(dvacc, 1.0f, dv_rs, dv_rowb);
#ifdef FA_KV_TIMING
        if (p.ws != nullptr && lane == 0) {
            const uint64_t t_end = __builtin_amdgcn_s_memtime(), rt_end = __builtin_amdgcn_s_memrealtime();
            float* out = p.ws + ((int64_t)blockIdx.x * 8 + wave) * 16;
            for (int i = 0; i < 7; ++i) out[i] = (float)tacc[i];
            out[7] = (float)n_iters;
            out[8] = (float)(t_loop - t_start);             // prologue
            out[9] = (float)(t_end - t_loop_end);           // epilogue
            out[10] = (float)(rt_start & 0xffffffull);       // wall clock, 100 MHz, low 24 bits
            out[11] = (float)(rt_end & 0xffffffull);
            out[12] = (float)tile; out[13] = (float)(batch * p.h_k + head_k);
            uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            uint32_t hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            out[14] = (float)(xcc & 0xf); out[15] = (float)((hwid >> 8) & 0xff);      // XCD, (CU / SE bits of HW_ID)
        }
#endif
        return;
    }
    // split head group: the UNSCALED fp32 sum goes to this split's plane of the workspace (a key's row is 4 * D bytes there)
    const int64_t plane = p.ws_rows * p.h_k * D;
    const int64_t row0 = (p.cu_seqlens_k != nullptr ? k_row0 : (int64_t)batch * p.seqlen_k) + n0;
    auto reduce_to_workspace = [&](f32x4 (&acc)[DB][2], int tensor) __attribute__((always_inline)) {
        hand_over(acc);
        if (qh == 0) {
            float* base = uniform_ptr(p.ws + ((int64_t)tensor * p.n_split + split) * plane + (row0 * p.h_k + head_k) * D);
            const rsrc_t rs = make_rsrc(base, (uint32_t)(keys_here - 1) * (uint32_t)(p.h_k * D * 4) + D * 4);
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    u32x4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = __builtin_bit_cast(uint32_t, acc[db][kc][r] + xch[(kb * XR + (db * 2 + kc) * 4 + r) * 64 + lane]);
                    buf_store16(rs, (uint32_t)(key_loc + 16 * kc) * (uint32_t)(p.h_k * D * 4) + (16 * db + 4 * g) * 4, w);      // rows >= keys_here fall outside the SRD
                }
        }
        __syncthreads();
    };
    reduce_to_workspace(dkacc, 0);
    reduce_to_workspace(dvacc, 1);
}

hipError_t launch_bwd_dkdv16(const BwdKernelParams& kp, int dtype, uint32_t grid, hipStream_t s) {
    if (kp.d == 64) {      // round 5: the same kernel at head_dim 64
        if (dtype == 0) {
            if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<_Float16, 64, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
            else hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<_Float16, 64, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        } else {
            if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<__bf16, 64, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
            else hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<__bf16, 64, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        }
        return hipGetLastError();
    }
    if (dtype == 0) {
        if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<_Float16, 128, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        else hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<_Float16, 128, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
    } else {
        if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<__bf16, 128, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        else hipLaunchKernelGGL((fa_bwd_dkdv16_kernel<__bf16, 128, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
    }
    return hipGetLastError();
}

}  // namespace fa
