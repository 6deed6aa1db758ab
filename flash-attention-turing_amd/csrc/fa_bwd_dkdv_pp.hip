// fa_bwd_dkdv_pp.hip — dK / dV kernel for head_dim 128 on MI355X (gfx950): two-group ping-pong.
//
//   dV = sum_i P_ij^T dO_i,  dK = scale * sum_i dS_ij^T Q_i          (reference flash_bwd_kernel.h:842-1676)
//
// Same tiling, register plan and arithmetic as the lock-step kernel in fa_bwd.hip (workgroup = 8 waves = 128 keys of one
// (batch, kv head); wave w owns key block w & 3 and the 32-row half qh = w >> 2 of every 64-row Q / dO tile; un-swapped layout,
// lane = key; dK^T / dV^T accumulators in AGPRs; K fragments in registers) - bit-identical results - with a different SCHEDULE:
//
//   a wave's tile is two phases,   P1 = S = Q K^T and dP = dO V^T          (16 MFMAs, operands by ds_read_b128)
//                                  P2 = P, dS (exp2, mask, multiply, round: ~90 VALU) then dV^T += dO^T P, dK^T += Q^T dS
//                                       (16 MFMAs, operands by transposing LDS reads),
//   each closed by a workgroup barrier, and the two q-half groups (waves 0-3 / 4-7; wave w and w + 4 share a SIMD) run ONE PHASE
//   APART for the whole life of the workgroup: while group A is in P1 (pure MFMA + LDS), group B is in P2, whose VALU block issues
//   under A's MFMAs and whose own MFMAs follow when A's are done.  In the lock-step kernel both waves of a SIMD reach the VALU block
//   together and the matrix pipe idles through it: 50 % MFMA utilisation at C4, 40 % of wave cycles parked (profiles/r2_shader_pmc_summary.json).
//
// Global "intervals" (barrier to barrier): A runs P1(t) in interval 2t and P2(t) in 2t+1; B runs P1(t) in 2t+1 and P2(t) in 2t+2.
// Q / dO tile t is therefore read in intervals 2t .. 2t+2, and the rings are 3 deep:
//   slot(t) = t % 3;  tile t+2 goes to the slot of tile t-1, last read in interval 2t (B's P2(t-1))
//   -> its LDS-DMA is ISSUED at the start of interval 2t+1 (A: head of P2(t), B: head of P1(t))
//   -> and AWAITED (vmcnt(0), then the barrier publishes it) at the end of interval 2t+2 (A: end of P1(t+1), B: end of P2(t)):
//      two full intervals in flight; first read in interval 2t+4.
//   The 64 LSE / D values of tile t+2 follow the same clock through one register of waves 0 / 1 (loaded at the head of A's P2(t),
//   transformed and stored to the 3-deep statistics ring at the end of A's P1(t+1)).
// LDS (D = 128): V[128 keys] 32 KiB | Q ring 3 x 16 KiB | dO ring 3 x 16 KiB | statistics 3 x 512 B = 129.5 KiB.  K never needs a
// home: it is staged once through ring slots 1-2 in the prologue and lives in registers from then on.
#include <type_traits>
#include "fa_bwd_dkdv_common.hpp"

namespace fa {

#ifndef FA_KVPP_PF1
#define FA_KVPP_PF1 4           // P1 steps (one MFMA each) whose LDS fragments are in flight
#endif
#ifndef FA_KVPP_PF2
#define FA_KVPP_PF2 4           // P2 steps likewise
#endif
#ifndef FA_KVPP_PRE_NL
#define FA_KVPP_PRE_NL 1        // request the -LSE block before the mid barrier (16 registers live across it)
#endif
#ifndef FA_KVPP_FR2_LATE
#define FA_KVPP_FR2_LATE 0      // 1: request the first transposed fragments only after P / dS are packed
#endif
#ifndef FA_KVPP_VREG
#define FA_KVPP_VREG 2          // V k-steps held in registers next to all 8 of K (see fa_bwd.hip: more spills)
#endif

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(kKvThreads, 2) void fa_bwd_dkdv_pp_kernel(const BwdKernelParams p) {
    constexpr int D = 128;
    constexpr int KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8, RING = 3;
    constexpr int KVB = kKvBlockN * ROWB;                   // the workgroup's K (or V) tile
    constexpr int TILEB = kKvBlockM * ROWB;                 // one Q (or dO) tile
    constexpr int STATB = 2 * kKvBlockM * 4;                // -lse*log2e + -D of one tile
    constexpr int OFF_Q = KVB, OFF_DO = KVB + RING * TILEB, OFF_STAT = KVB + 2 * RING * TILEB, OFF_KSTAGE = OFF_Q + TILEB;
    constexpr int LDS_BYTES = OFF_STAT + RING * STATB;
    static_assert(2 * TILEB == KVB, "K is staged through two ring slots");
    __shared__ __attribute__((aligned(1024))) char smem_raw[LDS_BYTES];     // (256-byte alignment is what the XOR addressing below relies on)
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* stat = smem + OFF_STAT;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave & 3, qh = wave >> 2;             // key block / q-half (= ping-pong group) of this wave

    int tile, batch, vhead, tiles_seq;
    if (!decode_work<kKvBlockN>(blockIdx.x, p.n_k_tiles, p.varlen_slots, p.cu_seqlens_k, p.b, p.h_k * p.n_split, tile, batch, vhead, tiles_seq)) return;
    const int head_k = vhead / p.n_split, split = vhead - head_k * p.n_split;
    const int heads_here = p.h_ratio / p.n_split;            // query heads of this workgroup (C ABI 3 head-group split, see fa_bwd.hip)

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch, v_boff = (int64_t)batch * p.v.batch,
            do_boff = (int64_t)batch * p.dout.batch, dk_boff = (int64_t)batch * p.dk.batch, dv_boff = (int64_t)batch * p.dv.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int qb = p.cu_seqlens_q[batch], kbeg = p.cu_seqlens_k[batch];
        sq = min(p.cu_seqlens_q[batch + 1] - qb, p.seqlen_q);   // clamp to the declared max_seqlen_q (padded LSE / D rows)
        sk = p.cu_seqlens_k[batch + 1] - kbeg;
        q_row0 = qb; k_row0 = kbeg;
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
    const int head_first = head_k * p.h_ratio + split * heads_here;

    const int key_row = kb * 32 + l31;                   // this lane's key inside the 128-key block
    const int wave_k_hi = n0 + kb * 32 + 31;

    // ---- LDS-DMA tables: a tile is 1 KiB pieces (64 lanes x 16 B, lane-linear in LDS, swizzle on the source offset); wave w moves
    // pieces [w*PPW, (w+1)*PPW).  Every LDS-DMA of this kernel is issued from inline asm (hipcc never sees one). --------------------
    constexpr int PPW_KV = (kKvBlockN * SLOTS / 64) / 8;  // pieces per wave: K/V tile (4)
    constexpr int PPW_Q = (kKvBlockM * SLOTS / 64) / 8;   //                  Q/dO tile (2)
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
    // LDS read addresses.  The XOR swizzle of the tile image (fa_device.hpp:lds_tile_off) only touches address bits 5-7, and so does
    // the k-step / d-block index, so ONE base register per access pattern serves every fragment:
    //   row reads  (8 contiguous d per lane; row l31, 16-byte slot 2*ks + hi):          row_base ^ (ks << 5)
    //   transposed reads (ds_read_b64_tr_b16; 4 x 16 block (sec, db)):                   (tr_base ^ ((db << 6) | (sec << 5))) + sec * 8 * ROWB
    // plus compile-time offsets (tensor, ring slot, 32-row block) in the instruction's immediate.  16 address registers become 2.
    const uint32_t row_base = lds_addr(smem) + lds_tile_off<D>(l31, hi);
    uint32_t tr_base;
    {
        const int L = lane & 15, g = (lane >> 4) & 1;
        tr_base = lds_addr(smem) + lds_tile_off<D>(4 * hi + (L >> 2), 2 * g + ((L & 3) >> 1)) + 8 * (L & 1);
    }
    auto row_read = [&](int ks, int imm) __attribute__((always_inline)) -> u32x4 {      // imm: byte offset of the 32-row block inside LDS
        return lds_read16((const FA_LDS char*)(uintptr_t)(row_base ^ (uint32_t)(ks << 5)), imm);
    };
    const float c = p.scale_log2e;

    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[db][r] = 0.f; dvacc[db][r] = 0.f; }

    // The Q / dO / statistics streams of ONE query head go through descriptors that are rebuilt only when the stream moves to the next
    // head of the group; inside a head a tile costs one scalar multiply and one VALU add per DMA piece (rebuilding them per tile was
    // ~110 SALU instructions = 600-1000 cycles per tile, profiles/r3_dkdv_phase_timing.log).
    srd_t q_srd = make_srd(nullptr, 0), do_srd = make_srd(nullptr, 0);
    rsrc_t st_rs = make_rsrc(nullptr, 0);
    auto set_head = [&](int hq) {
        const T* qb = uniform_ptr((const T*)p.q_ptr + q_boff + q_row0 * p.q.row + (int64_t)hq * p.q.head);
        const T* dob = uniform_ptr((const T*)p.do_ptr + do_boff + q_row0 * p.dout.row + (int64_t)hq * p.dout.head);
        q_srd = make_srd(qb, sq > 0 ? (uint32_t)(sq - 1) * q_rowb + ROWB : 0u);          // rows past the end of the sequence read zeros
        do_srd = make_srd(dob, sq > 0 ? (uint32_t)(sq - 1) * do_rowb + ROWB : 0u);
        // The 64 LSE / D values of a tile go through ONE register of waves 0 / 1 (wave 0: LSE -> -LSE*log2e, wave 1: D -> -D; other
        // waves: zero-record descriptor, no access), so the consumers need no per-element multiply / subtract: exp2(fma(s, c, nl)) and
        // a dP chain that starts from -D.  The load is unconditional (every wave, every iteration) and its register is consumed on
        // every path, so hipcc never has to guard it with a wait at a loop top (the round-1 stall, see fa_bwd.hip).
        const float* sb = uniform_ptr((wave == 0 ? p.lse_ptr : p.dsum_ptr) + ((int64_t)batch * p.h + hq) * p.lse_row_stride);
        st_rs = make_rsrc(sb, wave < 2 ? (uint32_t)sq * 4u : 0u);
    };
    // prefetch cursor: (query head, tile inside the head) of the next tile to request; advances one tile per call
    int pf_head = head_first, pf_tile = 0, pf_count = 0;
    auto pf_m0 = [&]() { return (qt_begin + pf_tile) * kKvBlockM; };
    auto pf_advance = [&]() {
        ++pf_count;
        if (++pf_tile == tiles_per_head) { pf_tile = 0; ++pf_head; if (pf_head < head_first + heads_here) set_head(pf_head); }
    };
    auto issue_tile = [&](int slot) {                      // tile at the cursor -> ring slot
        const uint32_t m0 = (uint32_t)pf_m0();
#pragma unroll
        for (int i = 0; i < PPW_Q; ++i) {
            const int piece = wave * PPW_Q + i;
            dma16_to_lds_hidden<false>(q_srd, q_src[i] + m0 * q_rowb, lds0 + OFF_Q + slot * TILEB + piece * 1024);
            dma16_to_lds_hidden<false>(do_srd, do_src[i] + m0 * do_rowb, lds0 + OFF_DO + slot * TILEB + piece * 1024);
        }
    };
    const float st_mult = wave == 0 ? -kLog2e : -1.0f;
    auto load_stat = [&](bool valid) -> float {            // !valid: an offset past every descriptor's range (returns 0, no access)
        const uint32_t off = valid ? ((uint32_t)pf_m0() + (uint32_t)lane) * 4u : 0xfffffff0u;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(st_rs, off, 0, 0));
    };
    auto store_stat = [&](float x, int slot) {
        *(FA_LDS float*)(stat + slot * STATB + wave * (kKvBlockM * 4) + lane * 4) = x * st_mult;
    };

    // ---- prologue: V -> its tile, K -> ring slots 1-2 (staging), Q(0) / dO(0) -> slot 0, statistics of tile 0 --------------------
#pragma unroll
    for (int i = 0; i < PPW_KV; ++i) {
        const int piece = wave * PPW_KV + i;
        dma16_to_lds_hidden<false>(k_srd, piece_src(piece, k_rowb), lds0 + OFF_KSTAGE + piece * 1024);
        dma16_to_lds_hidden<false>(v_srd, piece_src(piece, v_rowb), lds0 + piece * 1024);
    }
    if (n_iters > 0) {
        set_head(pf_head);
        issue_tile(0);
        if (wave < 2) store_stat(load_stat(true), 0);
        pf_advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // K (all 8 k-steps) and the first VREG k-steps of V stay in registers for the whole loop (the kernel is LDS-read-heavy: -5..-13 %)
    constexpr int VREG = FA_KVPP_VREG;
    u32x4 kreg[KS], vreg[VREG > 0 ? VREG : 1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kreg[ks] = row_read(ks, OFF_KSTAGE + kb * 32 * ROWB);
#pragma unroll
    for (int ks = 0; ks < VREG; ++ks) vreg[ks] = row_read(ks, kb * 32 * ROWB);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kreg[ks]));       // the reads have returned before the barrier that frees the staging slots
    __syncthreads();
    // tile 1 -> slot 1 (interval "-1"), awaited at the end of interval 0 like every odd-interval request
    float st_next = 0.f;
    {
        const bool more = n_iters > 1;
        if (more) issue_tile(1);
        st_next = load_stat(more);
        if (more) pf_advance();
    }
    if (qh == 1) {                                        // group B runs one interval behind group A from here on
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (its pieces of tile 1: this barrier closes interval 0)
        __syncthreads();
    }

    // ---- the tile loop -----------------------------------------------------------------------------------------------------------
    // One call of `do_tile` = both phases of one Q / dO tile with the ring slot as a compile-time constant (every LDS address is a base
    // register + immediate); the loop runs three tiles per trip.  Both phases are software-pipelined by hand and each one's first
    // operands are requested BEFORE the barrier that opens it (`pre-P1` at the end of P2, `pre-P2` at the end of P1): a phase starts on
    // an MFMA / a VALU instruction, not on an LDS round trip - alone on the matrix pipe (the partner wave is in its VALU block), a
    // wave with hipcc's one-step look-ahead ran its 16 MFMAs in ~750 cycles instead of 512 (profiles/r3_dkdv_phase_timing.log).
    int cur_tile = 0;                                     // tile-in-head index of the tile being computed
    f32x16 sacc, dpacc;                                   // S and dP - D of the current tile (live across the mid barrier)
    constexpr int NP1 = 2 * KS, PF1 = FA_KVPP_PF1;                  // P1 steps: j -> (ks = j >> 1, which = j & 1): 0 = S (Q rows), 1 = dP (dO rows [+ V rows])
    u32x4 fa1[PF1], fb1[PF1];                             // fragments of the first PF1 steps of the NEXT P1 (requested in pre-P1)
    f32x4 nl4[4];                                         // -LSE*log2e of this wave's 16 rows (requested in pre-P2)
    constexpr int NST = 4 * DB, PF2 = FA_KVPP_PF2;                  // P2 steps: j -> (half, db, which): which 0 -> dV (dO^T), 1 -> dK (Q^T)
    u32x4 fr2[PF2];
    const uint32_t st_row = (uint32_t)((32 * qh + 4 * hi) * 4);
    // (the wave's 32-row block inside a tile is a multiple of 256 bytes: folded into the bases, it commutes with the XOR)
    const uint32_t rowq_base = row_base + (uint32_t)(qh * 32 * ROWB), rowv_base = row_base + (uint32_t)(kb * 32 * ROWB),
                   trq_base = tr_base + (uint32_t)(qh * 32 * ROWB);
    // address = base ^ X, computed where it is used (one VALU; hipcc would hoist the loop-invariant XORs and hold all 16 results in
    // registers, which is exactly what this kernel cannot afford: it spilled K fragments instead)
    auto xor_now = [&](uint32_t base, auto xc) __attribute__((always_inline)) -> uint32_t {
        constexpr int X = decltype(xc)::value;
        if constexpr (X == 0) return base;
        uint32_t r;
        asm volatile("v_xor_b32 %0, %2, %1" : "=v"(r) : "v"(base), "n"(X));
        return r;
    };
    auto p1_frag_a = [&](auto jc, int slot) __attribute__((always_inline)) -> u32x4 {    // A operand of step j: Q rows (S) or dO rows (dP)
        constexpr int j = decltype(jc)::value;
        return lds_read16((const FA_LDS char*)(uintptr_t)xor_now(rowq_base, std::integral_constant<int, ((j >> 1) << 5)>{}), ((j & 1) ? OFF_DO : OFF_Q) + slot * TILEB);
    };
    auto p1_frag_b = [&](auto jc) __attribute__((always_inline)) -> u32x4 {              // B operand of a dP step whose V k-step is not in registers
        constexpr int j = decltype(jc)::value;
        return lds_read16((const FA_LDS char*)(uintptr_t)xor_now(rowv_base, std::integral_constant<int, ((j >> 1) << 5)>{}), 0);
    };
    auto pre_p1 = [&](int slot) __attribute__((always_inline)) {                         // -D block + the first PF1 fragments of tile in `slot`
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                  // dP chain starts from -D (registers 4*g4.. = rows 32*qh + 8*g4 + 4*hi + {0..3})
            const f32x4 nd4 = *(const FA_LDS f32x4*)(stat + slot * STATB + kKvBlockM * 4 + st_row + g4 * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) dpacc[4 * g4 + e] = nd4[e];
        }
        static_for<0, PF1>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            fa1[j] = p1_frag_a(jc, slot);
            if constexpr ((j & 1) && (j >> 1) >= VREG) fb1[j] = p1_frag_b(jc);
        });
    };
    auto pre_p2 = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) nl4[g4] = *(const FA_LDS f32x4*)(stat + slot * STATB + st_row + g4 * 32);
    };
    auto p2_frag = [&](auto jc, int slot) __attribute__((always_inline)) -> u32x4 {
        constexpr int j = decltype(jc)::value, half = j / (2 * DB), db = (j >> 1) % DB;
        const int imm = ((j & 1) ? OFF_Q : OFF_DO) + slot * TILEB + half * 16 * ROWB;
        const u32x2 a0 = lds_read_tr8((const FA_LDS char*)(uintptr_t)xor_now(trq_base, std::integral_constant<int, (db << 6)>{}), imm);
        const u32x2 a1 = lds_read_tr8((const FA_LDS char*)(uintptr_t)xor_now(trq_base, std::integral_constant<int, ((db << 6) | 32)>{}), imm + 8 * ROWB);
        return u32x4{a0.x, a0.y, a1.x, a1.y};
    };

    auto do_tile = [&](int it, auto slot_c) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value, SLOT1 = (SLOT + 1) % RING, SLOT2 = (SLOT + 2) % RING;
        const int m0 = (qt_begin + cur_tile) * kKvBlockM;
        const bool more1 = it + 1 < n_iters, more2 = pf_count < n_iters;      // tiles it+1 / it+2 exist (pf_count == it + 2 whenever the latter does)

        // ================= P1: S = Q K^T, dP - D = dO V^T - D =================
        if (qh == 1 && more2) issue_tile(SLOT2);          // B: head of P1 = start of an odd interval
        const int mh = m0 + 32 * qh;                       // first query row of this wave's half
        // causal mask, 2 VALU per element and only on diagonal tiles: element r = 4*g4 + e is query row mh + 8*g4 + 4*hi + e, visible
        // iff key <= row + delta  <=>  8*g4 + e >= thr.  No wave-level skip of fully masked wave-tiles, on purpose (a branch around
        // asm-accumulator MFMAs makes hipcc copy all 128 accumulators out and back, fa_bwd.hip): they run with every element masked.
        const bool need_mask = CAUSAL && (wave_k_hi > mh + delta);
        const int thr = (n0 + key_row) - (mh + 4 * hi + delta);
        {
            u32x4 fa[NP1], fb[NP1];
            static_for<0, PF1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                fa[j] = fa1[j];
                if constexpr ((j & 1) && (j >> 1) >= VREG) fb[j] = fb1[j];
            });
            static_for<0, NP1>([&](auto jc) {
                constexpr int j = decltype(jc)::value, ks = j >> 1;
                if constexpr (j + PF1 < NP1) {
                    fa[j + PF1] = p1_frag_a(std::integral_constant<int, j + PF1>{}, SLOT);
                    if constexpr (((j + PF1) & 1) && ((j + PF1) >> 1) >= VREG) fb[j + PF1] = p1_frag_b(std::integral_constant<int, j + PF1>{});
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((j & 1) == 0) {
                    if constexpr (ks == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
                    }
                    sacc = LP<T>::mfma(fa[j], kreg[ks], sacc);                                   // S = Q K^T  (rows = queries, lane = key)
                } else {
                    dpacc = LP<T>::mfma(fa[j], ks < VREG ? vreg[ks < VREG ? ks : 0] : fb[j], dpacc);    // dP - D = dO V^T - D
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if constexpr (FA_KVPP_PRE_NL) pre_p2(SLOT);
        if (qh == 0) {                                     // A: end of P1 = end of an even interval
            if (wave < 2 && more1) store_stat(st_next, SLOT1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile it+1 have landed
        }
        asm volatile("" :: "v"(st_next));
        __syncthreads();

        // ================= P2: P, dS; dV^T += dO^T P, dK^T += Q^T dS =================
        if (qh == 0 && more2) issue_tile(SLOT2);          // A: head of P2 = start of an odd interval
        st_next = load_stat(more2);                        // every wave, every iteration (B and waves 2-3: zero-record descriptor)
        if (more2) pf_advance();
        if constexpr (!FA_KVPP_PRE_NL) pre_p2(SLOT);
        f32x16 pacc;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int e = 0; e < 4; ++e) pacc[4 * g4 + e] = fast_exp2(__builtin_fmaf(sacc[4 * g4 + e], c, nl4[g4][e]));   // P = exp(s*scale - LSE) (flash_bwd_kernel.h:1339)
        // (the statistics registers are free again: request the first transposed fragments now, behind the rest of the VALU block)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FA_KVPP_FR2_LATE) static_for<0, PF2>([&](auto jc) { fr2[decltype(jc)::value] = p2_frag(jc, SLOT); });
        __builtin_amdgcn_sched_barrier(0);
        if (need_mask) {                                    // wave-uniform branch: only diagonal tiles pay for the select
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = (8 * (r >> 2) + (r & 3) >= thr) ? pacc[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = pacc[r] * dpacc[r];         // dS = P * (dP - D) (flash_bwd_kernel.h:1354)
        u32x4 pfr[2], dsfr[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            pfr[half] = pack_c_half<T>(pacc, half);             // P rounded (flash_bwd_kernel.h:1359)
            dsfr[half] = pack_c_half<T>(sacc, half);            // dS rounded (:1360)
        }
        if constexpr (FA_KVPP_FR2_LATE) {
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, PF2>([&](auto jc) { fr2[decltype(jc)::value] = p2_frag(jc, SLOT); });
        }
        {
            u32x4 fr[NST];
            static_for<0, PF2>([&](auto jc) { fr[decltype(jc)::value] = fr2[decltype(jc)::value]; });
            static_for<0, NST>([&](auto jc) {
                constexpr int j = decltype(jc)::value, half = j / (2 * DB), db = (j >> 1) % DB;
                if constexpr (j + PF2 < NST) fr[j + PF2] = p2_frag(std::integral_constant<int, j + PF2>{}, SLOT);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j & 1) LP<T>::mfma_agpr(dkacc[db], fr[j], dsfr[half]);      // dK^T += Q^T dS    (AGPR accumulator)
                else LP<T>::mfma_agpr(dvacc[db], fr[j], pfr[half]);                       // dV^T += dO^T P    (AGPR accumulator)
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if (more1) pre_p1(SLOT1);                          // tile it+1 has been resident for at least one interval
        if (qh == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // B: end of P2 = end of an even interval
        __syncthreads();
        if (++cur_tile == tiles_per_head) cur_tile = 0;
    };
    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, 1>;
    using s2 = std::integral_constant<int, 2>;
    if (n_iters > 0) pre_p1(0);                            // (tile 0 and its statistics were published by the prologue's first barrier)
    {
        int it = 0;
        for (; it + 3 <= n_iters; it += 3) {
            do_tile(it, s0{});
            do_tile(it + 1, s1{});
            do_tile(it + 2, s2{});
        }
        if (it < n_iters) do_tile(it, s0{});
        if (it + 1 < n_iters) do_tile(it + 1, s1{});
    }
    if (qh == 0) __syncthreads();                          // group A waits for B's last phase (equal barrier counts)

    dkdv_epilogue<T, D, OFF_STAT>(p, smem, dkacc, dvacc, batch, head_k, split, k_row0, n0, keys_here, dk_base, dv_base);
}

hipError_t launch_dkdv_pp(const BwdKernelParams& kp, int dtype, uint32_t grid, hipStream_t s) {
    if (dtype == 0) {
        if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv_pp_kernel<_Float16, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        else hipLaunchKernelGGL((fa_bwd_dkdv_pp_kernel<_Float16, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
    } else {
        if (kp.is_causal) hipLaunchKernelGGL((fa_bwd_dkdv_pp_kernel<__bf16, true>), dim3(grid), dim3(kKvThreads), 0, s, kp);
        else hipLaunchKernelGGL((fa_bwd_dkdv_pp_kernel<__bf16, false>), dim3(grid), dim3(kKvThreads), 0, s, kp);
    }
    return hipGetLastError();
}

}  // namespace fa
