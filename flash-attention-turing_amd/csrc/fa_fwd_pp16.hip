// fa_fwd_pp16.hip -- the forward kernel of fa_fwd_pp.hip (same workgroup shape, rings, LDS-DMA, two-group ping-pong, optimistic softmax,
// x3-unrolled loop; read its header first) re-tiled for v_mfma_f32_16x16x32_{f16,bf16}.
//
// Why: under matrix load this chip is power-limited, and the limit depends on the MFMA shape - a chip-wide loop of 16x16x32 MFMAs on N(0,1)
// operands sustains 1.98 PFLOP/s against 1.66 for 32x32x16 (half the accumulator traffic per FLOP), and still +12 % inside this kernel's
// own mix of 4 VALU + 1 KiB of LDS reads per 32768 FLOP (tools/powerbench, profiles/r3_powerbench_mfma_variants.log).
//
// What changes against the 32x32x16 kernel (everything else is identical, including the LDS tile image and every byte the DMA moves):
//   * a lane is (k-group g = lane >> 4, column n = lane & 15): it owns TWO query rows of its wave's 32 (n and n + 16) and, of every 16-key
//     score block, the 4 keys of its group; one LDS fragment (K rows / transposed V) feeds two MFMAs, one per query column, so LDS bytes
//     and VALU work per FLOP are unchanged;
//   * S^T = K Q^T in 16 x 16 blocks [key block][query column], contraction in 4 steps of 32 d; O^T = V^T P^T in 16-d blocks, contraction in
//     chunks of 32 keys: a lane's P values of key blocks 2c and 2c+1 ARE the 8 k-slots of chunk c (C layout -> B operand, no data movement);
//   * row max / row sum span four lanes (v_permlane16_swap + v_permlane32_swap) instead of two;
//   * two k-slot permutations (d chunks {0,3,1,2}, key sub-blocks {0,2,1,3}) keep the unchanged XOR-swizzled tile image bank-conflict free
//     under the new access patterns.
#include "fa_device.hpp"
#include "fa_params.hpp"

#include <type_traits>

namespace fa {

constexpr int kFwdThreads = 512;
constexpr float kPpDeferLog2 = 6.0f;
#define FA_PP_MIN_WAVES(D, BN) 2
// FA_PP16_FOLD_MAX (round 3: Q pre-scaled by log2(e)/sqrt(d), score chains started from -running_max; +5 % fp16) was measured and rejected
// for what rounding Q * scale does to LSE (profiles/r3_fwd_mfma16_ab.log); the code left the source in round 4 (history: c6d38fe).
//
// FA_PP16_MFMA_ROWSUM (round 4): the softmax row sum l leaves the VALU.  One extra MFMA per 32-key chunk and query column multiplies an
// all-ones A operand with the P^T fragment that feeds P.V anyway: every row of the 16 x 16 result is the chunk's column sum, already
// reduced over the four lane groups (+4 MFMAs on 64 per tile, -32 v_add_f32 per wave and tile).  The optimistic pass is then guarded by
// the largest packed P of the lane (v_pk_maximum3_f16 on the B-operand registers, 11 VALU) instead of the VALU partial sum.
// l is then the sum of the ROUNDED P values (fp32 accumulation in the matrix pipe): O = sum(round(P) V) / sum(round(P)) is a proper convex
// combination, and LSE carries the rounding noise of P: |dLSE| <= 2^-12 worst case (fp16), ~1.4e-4 / sqrt(effective keys) typical.
//   0 = off (VALU sums), 1 = fp16 only (default: bf16's 2^-9 would show in LSE on rows dominated by one key), 2 = both dtypes.
// FA_PP16_EXACT_TILES: the first tiles of every workgroup (and every masked / leftover tile) keep the exact VALU sums of the unrounded P.
// The rounding noise of a sum of N rounded terms is ~1.4e-4 * sqrt(e / N) of l for N(0,1) scores (2.9e-5 at N = 64, 7e-6 at N = 1024), so
// rows that see few keys - the first rows of a causal problem, short sequences - would carry 1e-4 in LSE; with 16 exact tiles (1024 keys) a
// row's MFMA-summed part always rides on >= 1024 exactly summed keys in front of it and the worst LSE deviation measured at the BASELINE
// sizes stays below 5e-5 (profiles/r4_lse_error_distribution.json).  Costs nothing where it matters: tiles 0..15 of 256 at 16k.
#ifndef FA_PP16_MFMA_ROWSUM
#define FA_PP16_MFMA_ROWSUM 1
#endif
// FA_PP16_ROLE_DMA (round 4): in the unrolled steady loop the waves of group A (0-3) move the K tiles and those of group B (4-7) the V tiles,
// four 1-KiB pieces each, instead of two pieces of both.  Why: with symmetric roles a tile requested in S(u) has to be complete when S(u) ends
// (group B runs one phase behind group A: its pieces of K(u+2) are read by group A two barriers later, and the first fragments of V(u+1) are
// prefetched one phase after their request) - one softmax phase, ~0.8 us, to cover an L2 miss.  With the roles split, K(u+2) requested in A's S(u)
// is first read in A's M(u+2), and V(u+2) requested in B's S(u) is first touched by group A's prefetch at the end of A's S(u+2): both can stay in
// flight for a whole tile period and are retired by a COUNTED wait at the end of the requesting group's NEXT softmax phase (step_c).  Requests
// are unconditional there (a tile past the sequence's end is a zero fill, one past the mask's end is fetched and never read).
// Invariants behind "unconditional" (ADVICE r4): (1) a ring slot that received a tile >= n_tiles is never read - the loops stop at n_tiles and the slot's next tenant
// is written before its first read; (2) the 32-bit source offset (t * BN) * row_bytes cannot wrap back into the descriptor's range: the loop runs only with
// n_main >= 4, i.e. seqlen_k >= 4 * BN, and the C-ABI refuses a sequence that spans 2^31 bytes (fa_capi.hip:check_tensor), so (seqlen_k + 2 * BN) * row_bytes
// <= 1.5 * seqlen_k * row_bytes < 2^32 and a request two tiles past the end lands beyond num_records (zero fill), never below it.
#ifndef FA_PP16_ROLE_DMA
#define FA_PP16_ROLE_DMA 1
#endif
// (Round 5, measured and not kept - code in the history at d4db12e: FA_PP16_ROWSUM_IN_S, the tile's row-sum MFMAs issued at the end of the softmax phase instead of riding
// in the wave's next matrix phase: bit-identical, +-0.3 %, profiles/r5_fwd_phase_balance_ablations.log.)
#ifndef FA_PP16_EXACT_TILES
#define FA_PP16_EXACT_TILES 16
#endif
#ifndef FA_PP16_EXACT_TILES_BF16
#define FA_PP16_EXACT_TILES_BF16 64      // (only with FA_PP16_MFMA_ROWSUM == 2)
#endif
#ifndef FA_PP16_ABL
#define FA_PP16_ABL 0       // timing-only ablations (results WRONG), bit mask: 1 the steady loop does not wait for its LDS-DMA, 2 does not issue it, 4 no exponentials (one multiply
#endif                      // per score instead of fma + exp), 8 no LDS fragment reads in the matrix phases (profiles/r4_fwd_pp16_ablations.log), 16 the softmax phase skips the second half of the
                            // tile's exponentials, 32 the next matrix phase carries them as extra work (profiles/r5_fwd_phase_balance_ablations.log)
// FA_PP16_DMA_DEBUG (round 5; test builds only, the product is 0 and its ISA does not change): adversarial timing for the LDS-DMA protocol of the
// unrolled steady loop.  A race of the class "a consumer reads a ring slot before the wait + barrier that publishes the producer's pieces" is invisible
// to every value test as long as the DMA is usually early (profiles/r4_fwd_counted_wait_racy_form_ab.log was bit-identical on every shape and wrong).
//   1 = LATE ISSUE: every request of the loop is issued at the last point the protocol itself allows - directly in front of the wait that retires it
//       (followed by vmcnt(0)): the bytes land as late as a correct protocol can tolerate, a wrong one reads the slot's previous tenant;
//   2 = SLEEPY GROUP: group B's waves sleep ~4000 cycles (more than a tile period) in front of every request.
// FA_PP16_RACY (test builds only, with FA_PP16_ROLE_DMA=0): the documented WRONG form - symmetric roles, K(u+2) left in flight across the barrier and
// retired by a counted wait at the end of the NEXT softmax phase - kept so that tests/test_dma_protocol_gpu.py can show the late-issue build catches it.
#ifndef FA_PP16_DMA_DEBUG
#define FA_PP16_DMA_DEBUG 0
#endif
#ifndef FA_PP16_RACY
#define FA_PP16_RACY 0
#endif
#if FA_PP16_RACY && FA_PP16_ROLE_DMA
#error "FA_PP16_RACY is the symmetric-role form: build with -DFA_PP16_ROLE_DMA=0"
#endif
// (Round 5, measured and not kept - code in the history at 1ef1631: FA_PP16_FOLD_ROWS, wave w owning the 16-row blocks w and 15 - w under a causal mask with per-column
// MFMA skipping on the diagonal: +0.6 % at 16k, +3..9 % below - the masked tiles are LDS / VALU bound and every wave then runs all four, profiles/r5_fwd_fold_rows_ab.log.)
#ifndef FA_PP16_DMA_FUSED
#define FA_PP16_DMA_FUSED 1    // (round 5) the four role pieces of a tile as one statement (fa_device.hpp:dma16x4_to_lds_hidden)
#endif
#ifndef FA_PP16_SKIP_PAD
#define FA_PP16_SKIP_PAD 1     // (round 5) the steady loop's softmax phase drops its wait-state pad when the fused DMA statement stands in front of it
#endif
// (Round 5, measured and not kept: a bare s_barrier instead of __syncthreads() at the end of the steady loop's softmax phase, so that hipcc does not wait (lgkmcnt(0)) for the
// fragments m_prefetch has just requested - bit-identical, no gain on top of the two switches above: the wave waits at that barrier anyway, profiles/r5_fwd_sphase_trim_ab.log.)
// (Round 6, measured and not kept (the switches were never committed; the logs say what was built): FA_PP16_DMA_STAGGER, the four waves of a group requesting their role pieces a quarter of the
// softmax pass apart instead of together behind the barrier: bit-identical, +9..41 % (the scalar branches inside the pass cost hipcc its schedule),
// profiles/r6_fwd_dma_stagger_ab.log; FA_PP16_PK_FMA, the multiply-subtract in front of every exponential two scores at a time (v_pk_fma_f32, 95 -> 79 VALU per wave and
// tile, bit-identical): +10..25 % - beside a partner wave that issues MFMAs a v_pk_fma_f32 takes 21 cycles where a v_fma_f32 takes 8 (tools/ubench, profiles/r6_ubench_cadence.log),
// profiles/r6_fwd_pk_fma_ab.log; a SIMPLE instance - four waves, 128 rows, two-slot rings with the staging aliased (64 KiB), every tile through iteration(), TWO workgroups per
// compute unit, for short sequences: value-correct on its first run and +3..+25 % slower from 512 to 4k (causal 512: -2..-5 %), profiles/r6_fwd_simple_two_per_cu_ab.log;
// after the phase stamps (FA_FWD_TIMING below: the two sides of a step are balanced at ~1320 cycles): FA_PP16_ROWSUM_DOT2, the MFMA-summed tiles' row sums by v_dot2c_f32_f16 in the
// softmax pass instead of four MFMAs per tile: +3..5 %, profiles/r6_fwd_rowsum_dot2_ab.log; FA_PP16_DMA_SPREAD, the four role pieces a quarter of the pass apart, branch-free: +9..56 %,
// profiles/r6_fwd_dma_spread_ab.log; FA_PP16_DMA_IN_M, the pieces between the MFMAs of the wave's own matrix phase (K(u+2) / V(u+1) in M(u)): bit-identical, +5..10 % - a matrix phase that
// issues VMEM stalls, profiles/r6_fwd_dma_in_m_ab.log; FA_PP16_ONE_BARRIER, one workgroup barrier per step (the instance between group A's matrix and softmax phase dropped, every slot hand-over
// still fenced): bit-identical, +0.4..6.6 % - the groups drift out of their alternation and the older one starves its partner, profiles/r6_fwd_one_barrier_ab.log; FA_PP16_S_PRIO, issue
// priority 1 / 2 during a wave's softmax side: +-1 %, head_dim 64 included, profiles/r6_fwd_softmax_prio_ab.log.)
#ifndef FA_PP16_PF
#define FA_PP16_PF 2        // LDS fragments in flight ahead of their MFMAs in a matrix phase (1-3 within 1 %, 2 best; 6: +1 %, 8: +2..4 %)
#endif

// FA_FWD_TIMING (round 6; timing builds only, the product does not define it and its ISA does not change): s_memtime stamps around the parts of a steady-loop step -
// matrix phase, the barrier behind it, LDS-DMA requests, softmax pass, fragment prefetch + counted DMA wait, the barrier behind that - summed per wave over the unrolled loop and left,
// with the step count, in the first 8 LSE entries of the wave's rows (LSE is WRONG in such a build; tools/phase_timing_fwd.py reads them, profiles/r6_fwd16_phase_timing.log)
#ifdef FA_FWD_TIMING
#define FA_FWD_STAMP(i) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc[i] += (uint32_t)(now_ - tlast); tlast = now_; } while (0)
#else
#define FA_FWD_STAMP(i) do { } while (0)
#endif
// QB = 16-row query columns per wave: 2 (256-row workgroups, the product) or 3 (384 rows; every K / V^T fragment then feeds three MFMAs - a third
// fewer LDS operand bytes per FLOP - with 32-key tiles so that S / P shrink enough for O (96) + Q (48) to stay at two waves per SIMD; round 5, FA_FWD_D128_QB)
template <typename T, int D, bool CAUSAL, int BN, int QB = 2>
__global__ __launch_bounds__(kFwdThreads, FA_PP_MIN_WAVES(D, BN)) void fa_fwd_pp16_kernel(const FwdKernelParams p) {
    constexpr int RW = 16 * QB, kFwdBlockM = 8 * RW;                          // query rows per wave / per workgroup
    constexpr int KS = D / 32, DB = D / 16, ROWB = D * 2, SLOTS = D / 8;      // QK^T k-steps of 32 d; 16-wide d blocks of O
    constexpr int kFwdBlockN = BN, NKB = BN / 16, NC = BN / 32;               // keys per tile; 16-key score blocks / 32-key P.V chunks per tile
    constexpr int TILEB = kFwdBlockN * ROWB;
    constexpr int RING = 3;
    constexpr int RINGB = 2 * RING * TILEB, STAGEB = kFwdBlockM * ROWB;     // D = 128: 96 KiB + 64 KiB = all 160 KiB of the CU
    constexpr bool ML = FA_PP16_MFMA_ROWSUM == 2 || (FA_PP16_MFMA_ROWSUM == 1 && std::is_same<T, _Float16>::value);      // row sums through the matrix pipe
    __shared__ __attribute__((aligned(16))) char smem_raw[RINGB + STAGEB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* kring = smem;
    FA_LDS char* vring = smem + RING * TILEB;
    FA_LDS char* stage = smem + RINGB;            // O block on its way out; every wave touches only its own 32 rows

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n16 = lane & 15;      // lane = (k-group g, column / row n16) of a 16x16x32 fragment
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;
    // k-slot permutations that keep the UNCHANGED LDS tile image (fa_device.hpp:lds_tile_off) bank-conflict free under the access patterns
    // of 16x16x32 fragments (checked exhaustively against the service groups of ds_read_b128 / ds_read_b64_tr_b16, profiles/NOTEBOOK.md 3d):
    //   d chunks:  lane group g contracts over d = 32*ks + 8*kPi[g] + 0..7          (A = K rows from LDS, B = Q^T from HBM: same permutation)
    //   keys:      row i = 4*gi + r of score block kb is key 16*kb + 4*kPi2[gi] + r   (A = K row i; C rows 4*g + r; P.V k-slots; V^T tr reads)
    const int pi_g = (0x2130 >> (4 * g)) & 3;             // kPi  = {0, 3, 1, 2}
    const int pi2_g = (0x3120 >> (4 * g)) & 3;            // kPi2 = {0, 2, 1, 3}
    const int q_row_a = wave * RW + n16;                  // this lane's QB query rows inside the workgroup's block: q_row_a + 16 * qb
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
#if FA_PP16_ROLE_DMA
    constexpr int RPW = BN * SLOTS / 256;     // pieces per wave and tile when four waves move a whole tile: 4
    const uint32_t r_rowb = group ? v_rowb : k_rowb;
    uint32_t dma_goff_r[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int chunk = ((wave & 3) * RPW + i) * 64 + lane, row = chunk / SLOTS, phys = chunk % SLOTS;
        dma_goff_r[i] = row * r_rowb + lds_tile_logical_slot<D>(row, phys) * 16;
    }
    const uint32_t lds_r0 = (group ? lds_addr(vring) : lds_addr(kring)) + (uint32_t)(wave & 3) * RPW * 1024;
#endif
    // K row reads (A of S^T = K Q^T): key block kb, k-step ks -> row 16*kb + 4*kPi2[gi] + r (i = n16 = 4*gi + r), 16-byte slot 4*ks + kPi[g]
    uint32_t k_rd[KS];
    {
        const int row = 4 * ((0x3120 >> (4 * (n16 >> 2))) & 3) + (n16 & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] = lds_addr(kring) + lds_tile_off<D>(row, 4 * ks + pi_g);
    }
    // V^T transposed reads (A of O^T += V^T P^T): 16-lane group g points at the 4 key rows 4*kPi2[g] .. +3 of a 16-key block, 16 d wide
    uint32_t v_rd[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) v_rd[db] = lds_addr(vring) + lds_tile_off<D>(4 * pi2_g + (n16 >> 2), 2 * db + ((n16 & 3) >> 1)) + 8 * (n16 & 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(k_rd[ks]));      // (opaque: every fragment read is base register + immediate)
#pragma unroll
    for (int db = 0; db < DB; ++db) asm volatile("" : "+v"(v_rd[db]));

    set_current(tile);
    if (n_tiles == 0) {
        // a block of rows that sees no key at all (causal with seqlen_q > seqlen_k, or an empty key sequence): O = 0, LSE = 0 (flash_fwd_kernel.h:718,767), written straight from
        // registers - no Q load, no K / V tile, no LDS, no barrier (round 4: such a workgroup took ~38 us through the full prologue and epilogue, profiles/r4_fwd_dead_rows_ab.log)
        const int rows_here = rows_of(tile);
        const rsrc_t o_rs = make_rsrc(uniform_ptr(o_ptr_of(tile)), (uint32_t)(rows_here - 1) * o_rowb + ROWB);
        constexpr int O_CHUNKS_DEAD = (RW * SLOTS) / 64;
#pragma unroll
        for (int i = 0; i < O_CHUNKS_DEAD; ++i) {
            const int chunk = lane + i * 64, row = wave * RW + chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, u32x4{0u, 0u, 0u, 0u});      // rows >= rows_here fall outside the SRD
        }
        if (lane < RW && wave * RW + lane < rows_here) lse_bh[tile * kFwdBlockM + wave * RW + lane] = 0.f;
        return;
    }

    // Q^T fragments (B operand), two query columns per lane: qf[ks][qb] = Q[q_row_a + 16*qb][32*ks + 8*kPi[g] .. +7]
    u32x4 qf[KS][QB];
    {
        const rsrc_t q_rs = make_rsrc(uniform_ptr(q_ptr_of(tile)), (uint32_t)(rows_of(tile) - 1) * q_rowb + ROWB);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) qf[ks][qb] = buf_load16(q_rs, (uint32_t)(q_row_a + 16 * qb) * q_rowb + (4 * ks + pi_g) * 16);
    }

    f32x4 oacc[DB][QB];                                   // O^T: d rows 16*db + 4*g + r, query column qb
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) oacc[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QB], l_run[QB];                                      // per query column; l_run is this lane's PARTIAL row sum (its 4 of every 16 keys)
    // ML: l as a 16 x 16 MFMA tile per query column, ones(16 x 32) * P^T: all four registers of every lane hold the column's FULL row sum
    f32x4 lacc[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m_run[qb] = kNegBig; l_run[qb] = 0.f; lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    u32x4 ones_a = {LP<T>::kOnes2, LP<T>::kOnes2, LP<T>::kOnes2, LP<T>::kOnes2};
    if constexpr (ML) asm volatile("" : "+v"(ones_a));           // (a register-resident constant: MFMA operands cannot be literals)

    int ring_u = 0, ring_um1 = 2, ring_up1 = 1;       // slot of tile u, of u-1 (== u+2), of u+1 in the 3-deep rings
    auto dma_k_tile = [&](const srd_t& srd, int t, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < (FA_PP16_DMA_FUSED && DPW == 2 ? 0 : DPW); ++i) dma16_to_lds_hidden<false>(srd, (uint32_t)(t * kFwdBlockN) * k_rowb + dma_goff_k[i], lds_k0 + slot * TILEB + i * 1024);
#if FA_PP16_DMA_FUSED
        if constexpr (DPW == 2) dma16x2_to_lds_hidden(srd, (uint32_t)(t * kFwdBlockN) * k_rowb, dma_goff_k[0], dma_goff_k[1], lds_k0 + slot * TILEB);
#endif
    };
    auto dma_v_tile = [&](const srd_t& srd, int t, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < (FA_PP16_DMA_FUSED && DPW == 2 ? 0 : DPW); ++i) dma16_to_lds_hidden<false>(srd, (uint32_t)(t * kFwdBlockN) * v_rowb + dma_goff_v[i], lds_v0 + slot * TILEB + i * 1024);
#if FA_PP16_DMA_FUSED
        if constexpr (DPW == 2) dma16x2_to_lds_hidden(srd, (uint32_t)(t * kFwdBlockN) * v_rowb, dma_goff_v[0], dma_goff_v[1], lds_v0 + slot * TILEB);
#endif
    };

#if FA_PP16_ROLE_DMA
    const srd_t r_srd = group ? v_srd : k_srd;
    auto dma_role_tile = [&](int t, int slot) __attribute__((always_inline)) {      // group A: K(t), group B: V(t)
#pragma unroll
        for (int i = 0; i < (FA_PP16_DMA_FUSED && (RPW == 4 || RPW == 2) ? 0 : RPW); ++i) dma16_to_lds_hidden<false>(r_srd, (uint32_t)(t * kFwdBlockN) * r_rowb + dma_goff_r[i], lds_r0 + slot * TILEB + i * 1024);
#if FA_PP16_DMA_FUSED
        if constexpr (RPW == 4) dma16x4_to_lds_hidden(r_srd, (uint32_t)(t * kFwdBlockN) * r_rowb, dma_goff_r[0], dma_goff_r[1], dma_goff_r[2], dma_goff_r[3], lds_r0 + slot * TILEB);
        if constexpr (RPW == 2) dma16x2_to_lds_hidden(r_srd, (uint32_t)(t * kFwdBlockN) * r_rowb, dma_goff_r[0], dma_goff_r[1], lds_r0 + slot * TILEB);      // (32-key tiles of the experiment build)
#endif
    };
#endif

    // ---- prologue: K(0), V(0), K(1) into the rings (past-the-end tiles arrive as zeros) ---------------
    if (n_tiles > 0) {
        dma_k_tile(k_srd, 0, 0);
        dma_v_tile(v_srd, 0, 0);
        dma_k_tile(k_srd, 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (group == 1) __syncthreads();          // group B runs one phase behind group A, for the whole life of the workgroup

    f32x4 sacc[NKB][QB];                              // S^T: key rows 4*g + r of score block kb (= keys 16*kb + 4*kPi2[g] + r), query column qb
    u32x4 pf[NC][QB];                                 // P^T as B operand of chunk c: k-slots = the lane's 4 keys of block 2c, then of block 2c+1
    const int wave_q_lo = m0 + wave * RW, wave_q_hi = wave_q_lo + RW - 1;

    // One matrix phase = NPV P.V fragments (V(u-1)) + NQK QK^T fragments (K(u)); every fragment feeds TWO MFMAs (the lane's two query columns),
    // so LDS bytes per FLOP are those of the 32x32x16 kernel.  Fragment j + PF is requested before the MFMAs of fragment j.
    constexpr int NPV = NC * DB, NQK = NKB * KS, NST = NPV + NQK, PF = FA_PP16_PF;
    auto m_frag = [&](int j, int slot_v, int slot_k) __attribute__((always_inline)) -> u32x4 {
#if FA_PP16_ABL & 8
        return qf[j % KS][(j >> 2) & 1];
#endif
        if (j < NPV) {
            const int db = j % DB, cch = j / DB;
            const u32x2 a0 = lds_read_tr8((const FA_LDS char*)(uintptr_t)v_rd[db], slot_v * TILEB + (32 * cch) * ROWB);
            const u32x2 a1 = lds_read_tr8((const FA_LDS char*)(uintptr_t)v_rd[db], slot_v * TILEB + (32 * cch + 16) * ROWB);
            return u32x4{a0.x, a0.y, a1.x, a1.y};
        }
        const int i = j - NPV, ks = i / NKB, kb = i % NKB;
        return lds_read16((const FA_LDS char*)(uintptr_t)k_rd[ks], slot_k * TILEB + 16 * kb * ROWB);
    };
    // `lsc`: the pending P tile was produced by the MFMA-sum path -> its row sums are taken here (compile-time yes / no, or a run-time bool)
    bool prev_ml = false;
    auto m_mfma = [&](auto jc, const u32x4& fr, auto lsc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NPV) {
            constexpr int db = j % DB, cch = j / DB;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) LP<T>::mfma16_acc(oacc[db][qb], fr, pf[cch][qb]);          // (this accumulator's previous MFMA is DB fragments = QB * DB MFMAs back)
            if constexpr (ML && db == DB - 1) {                      // the chunk's row sums (previous MFMA on lacc: a whole chunk back)
                bool take;
                if constexpr (std::is_same<decltype(lsc), bool>::value) take = lsc;
                else take = decltype(lsc)::value;
                if (take) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) LP<T>::mfma16_acc(lacc[qb], ones_a, pf[cch][qb]);
                }
            }
        } else {
            constexpr int i = j - NPV, ks = i / NKB, kb = i % NKB;
            if constexpr (ks == 0) {                               // (previous MFMA on this accumulator: NKB fragments back)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) LP<T>::mfma16_zero(sacc[kb][qb], fr, qf[ks][qb]);
            } else {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) LP<T>::mfma16_acc(sacc[kb][qb], fr, qf[ks][qb]);
            }
        }
    };
    u32x4 pre[PF];
    // full phase (P.V of the previous tile + QK^T of this one), ring slots as given (compile-time constants in the unrolled loop)
    // (the matrix phase at issue priority 1, so that the partner's softmax VALU yields to the MFMA stream: +-1 %, profiles/r4_prio_mfma_phase_ab.log)
    auto m_phase = [&](int slot_v, int slot_k, auto lsc) __attribute__((always_inline)) {
        u32x4 fr[NST];
        static_for<0, PF>([&](auto jc) { fr[decltype(jc)::value] = pre[decltype(jc)::value]; });
        static_for<0, NST>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + PF < NST) fr[j + PF] = m_frag(j + PF, slot_v, slot_k);
            __builtin_amdgcn_sched_barrier(0);
            m_mfma(jc, fr[j], lsc);
#if FA_PP16_ABL & 32
            if constexpr (!std::is_same<decltype(lsc), bool>::value) {
                if constexpr (decltype(lsc)::value && j < DB) {      // (timing only: two scores per query column behind each of the first DB fragment steps)
                    constexpr int kb = NKB / 2 + j / 4, r0 = (j % 4) & 2, qb = j & 1;
                    const float e0 = fast_exp2(__builtin_fmaf(sacc[kb][qb][r0], c, -m_run[qb] * c));
                    const float e1 = fast_exp2(__builtin_fmaf(sacc[kb][qb][r0 + 1], c, -m_run[qb] * c));
                    const uint32_t w = LP<T>::pack2(e0, e1);
                    asm volatile("" ::"v"(w));
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto m_prefetch = [&](int slot_v, int slot_k) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PF; ++j) pre[j] = m_frag(j, slot_v, slot_k);
    };
    // un-pipelined halves for the first / diagonal / last tiles
    auto pv_step = [&]() __attribute__((always_inline)) {
        static_for<0, NPV>([&](auto jc) { m_mfma(jc, m_frag(decltype(jc)::value, ring_um1, ring_u), prev_ml); });
    };
    auto qk_step = [&]() __attribute__((always_inline)) {
        static_for<NPV, NST>([&](auto jc) { m_mfma(jc, m_frag(decltype(jc)::value, ring_um1, ring_u), false); });
    };
    auto issue_dma_k = [&](int u) __attribute__((always_inline)) {
        if (u + 2 < n_tiles) dma_k_tile(k_srd, u + 2, ring_um1);
    };
    auto issue_dma_v = [&](int u) __attribute__((always_inline)) {
        if (u + 1 < n_tiles) dma_v_tile(v_srd, u + 1, ring_up1);
    };
    // max / sum over the four lane groups that share a query column (lanes l, l ^ 16, l ^ 32, l ^ 48)
    auto max4 = [&](float x) __attribute__((always_inline)) -> float {
        uint32_t u = __builtin_bit_cast(uint32_t, x);
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        x = fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
        return max_both_halves(x);
    };
    auto sum4 = [&](float x) __attribute__((always_inline)) -> float {
        uint32_t u = __builtin_bit_cast(uint32_t, x);
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        x = __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
        return sum_both_halves(x);
    };
    // `mlc`: this tile's row sums go through the matrix pipe (no VALU adds, packed-max guard) / stay exact in the VALU
    // `covered`: the caller has put >= 8 wait states of its own between the tile's last MFMA and this call (the steady loop: barrier + the fused DMA statement,
    // 18 of them) - the pad is then dropped, the register ties below stay (round 5; flash-attention-turing_amd/mfma_hazards.py counts the states in every instance's ISA)
    auto softmax_step = [&](int u, auto masked, auto maybe_first, auto mlc, auto covered) __attribute__((always_inline)) {
        constexpr bool MLT = ML && decltype(mlc)::value;
        // the scores were written by MFMAs issued from inline asm, which the hazard recogniser does not see: a 4-pass XDL write needs its
        // wait states before a VALU reads it (the barrier and the DMA issue in between usually cover them; this makes it unconditional)
        if constexpr (!decltype(covered)::value) asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
        // ... and the pad has to NAME the registers it protects: its "memory" clobber orders memory operations only, and hipcc is free to schedule a
        // register-only v_fma / v_exp that reads a score directly behind the MFMA that produces it, above the barrier and this pad (round 5, the 384-row
        // experiment build: v_fma_f32 two instructions behind its MFMA, P wrong by ~0.1; flash-attention-turing_amd/mfma_hazards.py walks the ISA of every instance for it).
        // Volatile asm statements keep their order, so nothing that reads sacc can move above these (no instructions are emitted).
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) asm volatile("" : "+v"(sacc[kb][qb]));
        const int n0 = u * kFwdBlockN;
        if constexpr (decltype(masked)::value) {
            const bool need_mask = (n0 + kFwdBlockN > sk) || (CAUSAL && (n0 + kFwdBlockN - 1 > wave_q_lo + delta));
            if (need_mask) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const int lim = (CAUSAL ? min(sk - 1, m0 + q_row_a + 16 * qb + delta) : sk - 1) - n0 - 4 * pi2_g;     // key = n0 + 16*kb + 4*kPi2[g] + r
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[kb][qb][r] = (16 * kb + r) <= lim ? sacc[kb][qb][r] : -INFINITY;
                }
            }
        }
        // One exponentiation pass against the running max AS IT STANDS (no row-max reduction): it stands if no lane's partial row sum exceeds
        // 2^kPpDeferLog2 - then no term does, and P, l and O stay in range.  Otherwise (Inf / NaN included; always on the first tile, whose
        // running max is still -1e30) the wave reduces the row maxima; if some row outgrew its running max by more than 2^kPpDeferLog2 the max is
        // refreshed, l and O are rescaled and the SAME pass runs once more (now every term is <= 1); if not, the pass already holds exactly
        // what the exact path would compute.
        // Tile 0 meets an empty running max: its first pass would always be thrown away, so the running max is seeded with the tile's row
        // maxima (no O or l to rescale yet) and the pass below stands at once.
        if constexpr (decltype(maybe_first)::value) {
            if (u == 0) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    float mx = sacc[0][qb][0];
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kb][qb][r]);
                    m_run[qb] = fmaxf(kNegBig, max4(mx));
                }
            }
        }
        float ps[QB];
        // exponentials of the tile against the running max as it stands -> P^T fragments; returns "some P above 2^kPpDeferLog2, or not finite"
        auto pass = [&]() __attribute__((always_inline)) -> bool {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float mc0 = m_run[qb] * c;
                ps[qb] = 0.f;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
#if FA_PP16_ABL & 16
                    if (MLT && kb >= NKB / 2) continue;      // (timing only: the second half of the tile's exponentials is not computed here)
#endif
#if FA_PP16_ABL & 4
                    const float p0 = sacc[kb][qb][0] * 0.01f + mc0 * 0.f, p1 = sacc[kb][qb][1] * 0.01f, p2 = sacc[kb][qb][2] * 0.01f, p3 = sacc[kb][qb][3] * 0.01f;
#else
                    const float p0 = fast_exp2(__builtin_fmaf(sacc[kb][qb][0], c, -mc0)), p1 = fast_exp2(__builtin_fmaf(sacc[kb][qb][1], c, -mc0));
                    const float p2 = fast_exp2(__builtin_fmaf(sacc[kb][qb][2], c, -mc0)), p3 = fast_exp2(__builtin_fmaf(sacc[kb][qb][3], c, -mc0));
#endif
                    if constexpr (!MLT) { ps[qb] += p0; ps[qb] += p1; ps[qb] += p2; ps[qb] += p3; }
                    if (kb & 1) { pf[kb >> 1][qb].z = LP<T>::pack2(p0, p1); pf[kb >> 1][qb].w = LP<T>::pack2(p2, p3); }
                    else { pf[kb >> 1][qb].x = LP<T>::pack2(p0, p1); pf[kb >> 1][qb].y = LP<T>::pack2(p2, p3); }
                }
            }
            if constexpr (MLT) {
                // largest packed P of the lane, both query columns: positive fp16 / bf16 bit patterns order like unsigned integers, and
                // v_pk_maximum3_f16 (IEEE maximum: NaN wins) on bf16 bits is monotone as long as they read as finite fp16, i.e. below 2^121;
                // anything above, Inf and NaN come out as a pattern above kBits64 as well.
                uint32_t t0, t1;
                if constexpr (QB != 2) {                              // any column count: one chain per query column, merged at the end
                    uint32_t tq[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        tq[qb] = pk_max3_f16_bits(pf[0][qb].x, pf[0][qb].y, pf[0][qb].z);
#pragma unroll
                        for (int cch = 1; cch < NC; ++cch) {
                            tq[qb] = pk_max3_f16_bits(tq[qb], pf[cch - 1][qb].w, pf[cch][qb].x);
                            tq[qb] = pk_max3_f16_bits(tq[qb], pf[cch][qb].y, pf[cch][qb].z);
                        }
                    }
                    static_assert(QB == 2 || QB == 3, "guard merge written for three columns");
                    t0 = pk_max3_f16_bits(tq[0], tq[1], tq[QB - 1]);
                    t0 = pk_max3_f16_bits(t0, pf[NC - 1][0].w, pf[NC - 1][1].w);
                    t0 = pk_max3_f16_bits(t0, pf[NC - 1][QB - 1].w, pf[NC - 1][QB - 1].w);
                    (void)t1;
                } else if constexpr (NC == 2) {
                    t0 = pk_max3_f16_bits(pf[0][0].x, pf[0][0].y, pf[0][0].z); t1 = pk_max3_f16_bits(pf[0][1].x, pf[0][1].y, pf[0][1].z);
                    t0 = pk_max3_f16_bits(t0, pf[0][0].w, pf[1][0].x); t1 = pk_max3_f16_bits(t1, pf[0][1].w, pf[1][1].x);
                    t0 = pk_max3_f16_bits(t0, pf[1][0].y, pf[1][0].z); t1 = pk_max3_f16_bits(t1, pf[1][1].y, pf[1][1].z);
                    t0 = pk_max3_f16_bits(t0, pf[1][0].w, t1);
                    t0 = pk_max3_f16_bits(t0, pf[1][1].w, pf[1][1].w);
                } else {                                              // any chunk count: one chain per query column, two words a step
                    t0 = pf[0][0].x; t1 = pf[0][1].x;
                    t0 = pk_max3_f16_bits(t0, pf[0][0].y, pf[0][0].z); t1 = pk_max3_f16_bits(t1, pf[0][1].y, pf[0][1].z);
#pragma unroll
                    for (int cch = 1; cch < NC; ++cch) {
                        t0 = pk_max3_f16_bits(t0, pf[cch - 1][0].w, pf[cch][0].x); t1 = pk_max3_f16_bits(t1, pf[cch - 1][1].w, pf[cch][1].x);
                        t0 = pk_max3_f16_bits(t0, pf[cch][0].y, pf[cch][0].z); t1 = pk_max3_f16_bits(t1, pf[cch][1].y, pf[cch][1].z);
                    }
                    t0 = pk_max3_f16_bits(t0, pf[NC - 1][0].w, pf[NC - 1][1].w);
                    t0 = pk_max3_f16_bits(t0, t1, t1);
                }
                const uint32_t both = max(t0, t0 << 16);          // top half = the larger of the two packed values
                return both > ((LP<T>::kBits64 << 16) | 0xffffu);
            } else {
                bool ok = true;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) ok = ok && ps[qb] <= 64.0f;
                return !ok;
            }
        };
        if constexpr (MLT) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pass()) != 0, 0)) {
                float mx[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    mx[qb] = sacc[0][qb][0];
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx[qb] = fmaxf(mx[qb], sacc[kb][qb][r]);
                    mx[qb] = max4(mx[qb]);
                }
                // (one decision for both query columns: refreshing a column that did not need it is exact too; a NaN / Inf score fails
                // both tests and the pass stands with its NaN)
                bool grew = false;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) grew = grew || (mx[qb] - m_run[qb]) * c > kPpDeferLog2;
                if (__builtin_amdgcn_ballot_w64(grew) != 0) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const float m_new = fmaxf(m_run[qb], mx[qb]);
                        const float alpha = fast_exp2((m_run[qb] - m_new) * c);
                        m_run[qb] = m_new;
                        l_run[qb] *= alpha;                       // (the exactly summed tiles of this workgroup)
#pragma unroll
                        for (int r = 0; r < 4; ++r) lacc[qb][r] *= alpha;
#pragma unroll
                        for (int db = 0; db < DB; ++db)
#pragma unroll
                            for (int r = 0; r < 4; ++r) oacc[db][qb][r] *= alpha;
                    }
                    (void)pass();
                }
            }
            return;
        }
        for (int attempt = 0;; ++attempt) {
            const bool over = pass();
            if (attempt != 0 || __builtin_amdgcn_ballot_w64(over) == 0) break;
            float mx[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                mx[qb] = sacc[0][qb][0];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx[qb] = fmaxf(mx[qb], sacc[kb][qb][r]);
                mx[qb] = max4(mx[qb]);
            }
            // (one decision for both query columns: refreshing a column that did not need it is exact too)
            bool grew = false;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) grew = grew || (mx[qb] - m_run[qb]) * c > kPpDeferLog2;
            if (__builtin_amdgcn_ballot_w64(grew) == 0) break;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float m_new = fmaxf(m_run[qb], mx[qb]);
                const float alpha = fast_exp2((m_run[qb] - m_new) * c);
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
                if constexpr (ML) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) lacc[qb][r] *= alpha;
                }
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[db][qb][r] *= alpha;
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) l_run[qb] += ps[qb];
    };
    auto advance_ring = [&]() __attribute__((always_inline)) {
        ring_um1 = ring_u;
        ring_u = ring_up1;
        ring_up1 = ring_up1 == 2 ? 0 : ring_up1 + 1;
    };
    auto end_s_phase = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // Epilogue, wave-local: normalise, round, stage this wave's 32 rows in LDS (own region), store them as whole rows.
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");        // (asm-issued MFMAs: results must have landed before the VALU below reads them)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) asm volatile("" : "+v"(oacc[db][qb]));
        const int rows_here = rows_of(t);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if constexpr (ML) asm volatile("" : "+v"(lacc[qb]));
            const float l_tot = ML ? sum4(l_run[qb]) + lacc[qb][0] : sum4(l_run[qb]);      // exactly summed tiles + MFMA-summed tiles
            // dead rows (row sum exactly 0): O = 0, LSE = 0; a NaN row sum is NOT dead, it propagates (flash_fwd_kernel.h:718,767: `!= 0`)
            const float inv = l_tot != 0.f ? fast_rcp(l_tot) : 0.f;
            const float lse = l_tot != 0.f ? (m_run[qb] * c + fast_log2(l_tot)) * kLn2 : 0.f;
            const int row = q_row_a + 16 * qb;
            if (g == 0 && row < rows_here) lse_bh[t * kFwdBlockM + row] = lse;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                u32x2 w;
                w.x = LP<T>::pack2(oacc[db][qb][0] * inv, oacc[db][qb][1] * inv);
                w.y = LP<T>::pack2(oacc[db][qb][2] * inv, oacc[db][qb][3] * inv);
                lds_write8(stage, lds_tile_off<D>(row, 2 * db + (g >> 1)) + 8 * (g & 1), w);      // d = 16*db + 4*g + {0..3}
            }
        }
        const rsrc_t o_rs = make_rsrc(uniform_ptr(o_ptr_of(t)), (uint32_t)(rows_here - 1) * o_rowb + ROWB);
        constexpr int O_CHUNKS = (RW * SLOTS) / 64;
#pragma unroll
        for (int i = 0; i < O_CHUNKS; ++i) {
            const int chunk = lane + i * 64, row = wave * RW + chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, lds_read16(stage, lds_tile_off<D>(row, slot)));   // rows >= rows_here fall outside the SRD
        }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    bool prev_active = false;                         // does this wave hold a P tile whose P.V is pending?
    auto iteration = [&](int u, auto masked) __attribute__((always_inline)) {
        bool active = true;
        if constexpr (decltype(masked)::value) active = !CAUSAL || (u * kFwdBlockN <= wave_q_hi + delta);
        if (prev_active) pv_step();
        if (active) qk_step();
        issue_dma_k(u);
        __syncthreads();
        issue_dma_v(u);
        if (active) softmax_step(u, masked, yes{}, no{}, no{});      // exact row sums in every masked / first tile
        prev_active = active;
        prev_ml = false;
        end_s_phase();
        advance_ring();
    };

#ifdef FA_FWD_TIMING
    uint32_t tacc[6] = {0u, 0u, 0u, 0u, 0u, 0u}, tsteps = 0u;
    uint64_t tlast = 0;
#endif
    int u = 0;
    if (n_main > 0) {
        iteration(0, no{});
        if (n_main > 1) m_prefetch(ring_um1, ring_u);
        // lsc: the pending P (tile uu - 1) has its row sums taken by MFMA in this step's matrix phase; mlc: tile uu's softmax leaves them to the next one
        auto step_c = [&](int uu, auto um1, auto u0, auto up1, auto lsc, auto mlc) __attribute__((always_inline)) {
            constexpr int S_UM1 = decltype(um1)::value, S_U = decltype(u0)::value, S_UP1 = decltype(up1)::value;
            m_phase(S_UM1, S_U, lsc);
            FA_FWD_STAMP(0);
            __syncthreads();
            FA_FWD_STAMP(1);
#if FA_PP16_ROLE_DMA
            // group A: K(u+2) -> the slot K(u-1) left (last read in M(u-1), which group B finished one barrier ago);
            // group B: V(u+2) -> the slot V(u-1) left (last read in M(u), which group B itself has just finished and group A one phase earlier).
            // Retired: the tile requested one softmax phase ago (K(u+1) / V(u+1)), by count - this phase's four pieces stay in flight.
            // The requests stand BEHIND the barrier on purpose: a wave moves pieces of the whole tile, and the other waves of its own group may still be inside
            // M(u) reading V(u-1) when the first one gets here - issued in front of the barrier they raced on one shape (round 5, profiles/r5_fwd_sphase_trim_ab.log (d)).
#if FA_PP16_DMA_DEBUG == 1
            // LATE ISSUE: the request the counted wait of this phase would retire (tile u+1, made one phase ago in the product) goes out here instead
            softmax_step(uu, no{}, no{}, mlc, no{});
            m_prefetch(S_U, S_UP1);
            dma_role_tile(uu + 1, S_UP1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#else
#if FA_PP16_DMA_DEBUG == 2
            if (group == 1) asm volatile("s_sleep 64" ::: "memory");
#endif
#if !(FA_PP16_ABL & 2)
            dma_role_tile(uu + 2, S_UM1);
#endif
            FA_FWD_STAMP(2);
            softmax_step(uu, no{}, no{}, mlc, std::integral_constant<bool, FA_PP16_DMA_FUSED && !(FA_PP16_ABL & 2) && FA_PP16_SKIP_PAD && (BN * (D / 8) / 256 == 4 || BN * (D / 8) / 256 == 2)>{});      // (12 wait states with the two-piece statement)
            FA_FWD_STAMP(3);
            m_prefetch(S_U, S_UP1);
#if !(FA_PP16_ABL & 3)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW) : "memory");
#endif
            FA_FWD_STAMP(4);
            __syncthreads();
            FA_FWD_STAMP(5);
#ifdef FA_FWD_TIMING
            ++tsteps;
#endif
#endif
#elif FA_PP16_RACY
            // the WRONG form (see the header): V(u+1) first so that the K(u+2) pieces are the youngest; the counted wait retires V(u+1) and the K(u+1)
            // pieces of the PREVIOUS phase and leaves K(u+2) in flight across the barrier - but the other group's half of K(u+2) is then retired
            // one barrier after this group starts reading the tile
#if FA_PP16_DMA_DEBUG == 1
            softmax_step(uu, no{}, no{}, mlc, no{});
            m_prefetch(S_U, S_UP1);
            dma_v_tile(v_srd, uu + 1, S_UP1);
            dma_k_tile(k_srd, uu + 1, S_UP1);      // (late issue: the K request this phase's counted wait retires)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#else
            dma_v_tile(v_srd, uu + 1, S_UP1);
            dma_k_tile(k_srd, uu + 2, S_UM1);
            softmax_step(uu, no{}, no{}, mlc, no{});
            m_prefetch(S_U, S_UP1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
            __syncthreads();
#endif
#else
            if (uu + 2 < n_tiles) dma_k_tile(k_srd, uu + 2, S_UM1);
            if (uu + 1 < n_tiles) dma_v_tile(v_srd, uu + 1, S_UP1);
            softmax_step(uu, no{}, no{}, mlc, no{});
            m_prefetch(S_U, S_UP1);
            end_s_phase();
#endif
        };
        using i0 = std::integral_constant<int, 0>;
        using i1 = std::integral_constant<int, 1>;
        using i2 = std::integral_constant<int, 2>;
        u = 1;
#ifdef FA_FWD_TIMING
        tlast = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#if FA_PP16_ROLE_DMA
        // entering the role-split loop: K(2) was requested by tile 0's iteration, V(2) by nobody yet (the symmetric scheme runs V one tile behind K)
        const bool role_loop = n_main >= 4;
        if (role_loop && group == 1) dma_role_tile(2, 2);
#endif
        // ring slot of tile u is u % 3: three steps per trip make every slot a constant
        // (counted in 64-key tiles: the same 1024 keys for any tile width; bf16, whose P rounds to 8 bits, rides on a longer exact prefix)
        constexpr int kExactTiles = (std::is_same<T, _Float16>::value ? FA_PP16_EXACT_TILES : FA_PP16_EXACT_TILES_BF16) * 64 / BN;
        constexpr int kExact = ML ? 1 + 3 * ((kExactTiles + 1) / 3) : 0;      // first MFMA-summed tile: start of a trip (16 for the default)
        if constexpr (ML) {
            for (; u + 3 <= n_main && u + 3 <= kExact; u += 3) {      // exactly summed tiles
                step_c(u, i0{}, i1{}, i2{}, no{}, no{});
                step_c(u + 1, i1{}, i2{}, i0{}, no{}, no{});
                step_c(u + 2, i2{}, i0{}, i1{}, no{}, no{});
            }
            if (u == kExact && u + 3 <= n_main) {
                step_c(u, i0{}, i1{}, i2{}, no{}, yes{});             // the pending P is still an exactly summed one
                step_c(u + 1, i1{}, i2{}, i0{}, yes{}, yes{});
                step_c(u + 2, i2{}, i0{}, i1{}, yes{}, yes{});
                u += 3;
                for (; u + 3 <= n_main; u += 3) {
                    step_c(u, i0{}, i1{}, i2{}, yes{}, yes{});
                    step_c(u + 1, i1{}, i2{}, i0{}, yes{}, yes{});
                    step_c(u + 2, i2{}, i0{}, i1{}, yes{}, yes{});
                }
                prev_ml = true;
            }
        } else {
            for (; u + 3 <= n_main; u += 3) {
                step_c(u, i0{}, i1{}, i2{}, no{}, no{});
                step_c(u + 1, i1{}, i2{}, i0{}, no{}, no{});
                step_c(u + 2, i2{}, i0{}, i1{}, no{}, no{});
            }
        }
#if FA_PP16_ROLE_DMA
        // leaving it: this wave's last request (one tile ahead of what the symmetric code below assumes; that code requests the tile again, the
        // same bytes into the same slot) is retired here; the next barrier publishes it long before its first read
#if FA_PP16_DMA_DEBUG == 1
        if (role_loop && u > 1) dma_role_tile(u + 1, ring_up1);      // (late issue: the product requested this tile in the loop's last step)
#endif
        if (role_loop) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif FA_PP16_RACY
#if FA_PP16_DMA_DEBUG == 1
        if (u > 1) dma_k_tile(k_srd, u + 1, ring_up1);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        for (; u < n_main; ++u) {                 // the last one or two steady-state tiles
            m_phase(ring_um1, ring_u, prev_ml);
            __syncthreads();
            issue_dma_k(u);
            issue_dma_v(u);
            softmax_step(u, no{}, no{}, no{}, no{});
            prev_ml = false;
            m_prefetch(ring_u, ring_up1);
            end_s_phase();
            advance_ring();
        }
    }
    for (; u < n_tiles; ++u) iteration(u, yes{});        // diagonal / ragged tiles
    if (prev_active) pv_step();
    epilogue(tile);
#ifdef FA_FWD_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) lse_bh[tile * kFwdBlockM + wave * RW + i] = (float)tacc[i];
        lse_bh[tile * kFwdBlockM + wave * RW + 6] = (float)tsteps;
        lse_bh[tile * kFwdBlockM + wave * RW + 7] = (float)tile;
    }
#endif
    if (group == 0) __syncthreads();          // group A waits for B's last phase (equal barrier counts)
}

// FA_FWD_D128_QB / FA_FWD_D128_BN (round 5): the head_dim-128 instance - query columns per wave (2: 256-row workgroups; 3: 384 rows) and keys per tile.
// The product is 2 / 64; 3 / 32 is the "three query columns" experiment (profiles/r5_fwd_qb3_ab.log), 2 / 32 isolates what the shorter phases cost.
#ifndef FA_FWD_D128_QB
#define FA_FWD_D128_QB 2
#endif
#ifndef FA_FWD_D128_BN
#define FA_FWD_D128_BN 64
#endif
int fwd_pp16_block_m(int d) { return d == 128 ? 128 * FA_FWD_D128_QB : 256; }      // query rows per workgroup: launch_fwd sizes the grid with it

hipError_t launch_fwd_pp16(const FwdKernelParams& kp, int dtype, uint32_t grid, hipStream_t stream) {
    if (grid == 0) return hipSuccess;
    if (kp.d == 64) {      // 128-key tiles: the same 16 KiB tile images, one workgroup per CU
        if (dtype == 0) {
            if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_pp16_kernel<_Float16, 64, true, 128>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
            else hipLaunchKernelGGL((fa_fwd_pp16_kernel<_Float16, 64, false, 128>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
        } else {
            if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_pp16_kernel<__bf16, 64, true, 128>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
            else hipLaunchKernelGGL((fa_fwd_pp16_kernel<__bf16, 64, false, 128>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
        }
        return hipGetLastError();
    }
    if (dtype == 0) {
        if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_pp16_kernel<_Float16, 128, true, FA_FWD_D128_BN, FA_FWD_D128_QB>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
        else hipLaunchKernelGGL((fa_fwd_pp16_kernel<_Float16, 128, false, FA_FWD_D128_BN, FA_FWD_D128_QB>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
    } else {
        if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_pp16_kernel<__bf16, 128, true, FA_FWD_D128_BN, FA_FWD_D128_QB>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
        else hipLaunchKernelGGL((fa_fwd_pp16_kernel<__bf16, 128, false, FA_FWD_D128_BN, FA_FWD_D128_QB>), dim3(grid), dim3(kFwdThreads), 0, stream, kp);
    }
    return hipGetLastError();
}

}  // namespace fa
