// fa_fwd_w4.hip -- head_dim-128 forward with ONE wave per SIMD: 4 waves x 64 query rows, the whole 512-register file per wave.
//
// Why (round 4): the 8-wave kernels (fa_fwd_pp.hip / fa_fwd_pp16.hip) pair a matrix-only wave with a softmax-only wave on every SIMD.  Measured
// (tools/ubench, profiles/r1_ubench*.log): a wave that issues only MFMAs leaves its SIMD partner ONE VALU slot per MFMA, while a wave that
// interleaves its OWN VALU work between its own MFMAs gets 4-6 of them nearly free; and every K / V fragment read from LDS feeds only two
// MFMAs there.  Here a wave owns 64 query rows, so
//   * every LDS fragment feeds FOUR MFMAs (half the LDS bytes per FLOP),
//   * the softmax of one half of the query columns is woven, instruction by instruction, into the QK^T MFMAs of the next tile and the
//     softmax of the other half into the P.V MFMAs (no partner wave, one workgroup barrier per tile instead of two),
//   * O^T (128 registers), Q^T (64) and the row sums live in the accumulation half of the register file and are touched by MFMAs only:
//     the asm statements below name a[0:211] directly ("asm-owned"; hipcc itself never allocates an AGPR in this translation unit:
//     no builtin MFMA, -amdgpu-mfma-vgpr-form, spilling to AGPRs switched off in build.py, and tests/test_kernel_resources_cpu.py checks that no
//     v_accvgpr instruction exists outside the asm statements).
//
// Tile schedule (tile = 64 keys; lane = (k-group g, column n16) and every index permutation exactly as in fa_fwd_pp16.hip, same LDS image):
//   step(u) = barrier | A(u): 64 MFMAs  S(u+1)^T = K(u+1) Q^T      woven with softmax of S(u),   query columns 2, 3 -> P(u)[2,3]
//                     | B(u): 64 MFMAs  O^T += V(u)^T P(u)^T (+ 8 ones-row MFMAs for the row sums)
//                                                                  woven with softmax of S(u+1), query columns 0, 1 -> P(u+1)[0,1]
//   The split is by QUERY column, not by key: a running-max refresh touches one column's O / l / P only, so nothing of that column is ever
//   "pending" at the two points where the optimistic pass is checked (end of A for columns 2, 3; end of B, when every P.V MFMA of the tile has
//   been issued, for columns 0, 1).  S of columns 2, 3 and P of columns 0, 1 are double-buffered by tile parity (x2 unrolled loop).
//   K / V tiles arrive by LDS-DMA into 3-deep rings, issued two tiles ahead (K(u+3), V(u+2) during step u), one 1-KiB piece at a time
//   between MFMAs, retired by a COUNTED s_waitcnt vmcnt(8) in front of the barrier (past-the-end tiles are requested too: the SRD range check
//   turns them into zero fills without memory traffic, so the count never changes).
#include "fa_device.hpp"
#include "fa_params.hpp"

#include <type_traits>

namespace fa {

constexpr int kW4Threads = 256;
constexpr int kW4BlockM = 256;
constexpr float kW4DeferLog2 = 6.0f;

// accumulation-register map (asm-owned)
constexpr int kAccO = 0;        // O^T[db][qb]   a[16 * db + 4 * qb .. + 3],   db = 0..7 (16 d each), qb = 0..3 (16 query rows each)
constexpr int kAccQ = 128;      // Q^T[ks][qb]   a[128 + 16 * ks + 4 * qb .. + 3]
constexpr int kAccL = 192;      // l[qb]         a[192 + 4 * qb .. + 3]   (ones(16 x 32) * P^T: every register holds the column's row sum)
constexpr int kAccOnes = 208;   // a[208:211]    packed 1.0

template <int R>
FA_DEV void acc_write(uint32_t v) { asm volatile("v_accvgpr_write_b32 a[%0], %1" ::"n"(R), "v"(v)); }
template <int R>
FA_DEV float acc_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R));
    return v;
}
template <int R>
FA_DEV void acc_scale(float f) {      // a[R] *= f   (rare path: running-max refresh)
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%1], %0" : "=&v"(t) : "n"(R), "v"(f));
}

template <typename T>
struct W4;
#define FA_W4_ASM(TYPE, SUF)                                                                                                                       \
    template <>                                                                                                                                   \
    struct W4<TYPE> {                                                                                                                             \
        template <int Q0>                                                                                                                         \
        static FA_DEV void qk_zero(f32x4& s, const u32x4& k) {                                                                                     \
            asm volatile("v_mfma_f32_16x16x32_" SUF " %0, %1, a[%2:%3], 0" : "=&v"(s) : "v"(k), "n"(Q0), "n"(Q0 + 3));                             \
        }                                                                                                                                         \
        template <int Q0>                                                                                                                         \
        static FA_DEV void qk_acc(f32x4& s, const u32x4& k) {                                                                                      \
            asm volatile("v_mfma_f32_16x16x32_" SUF " %0, %1, a[%2:%3], %0" : "+v"(s) : "v"(k), "n"(Q0), "n"(Q0 + 3));                              \
        }                                                                                                                                         \
        template <int O0>                                                                                                                         \
        static FA_DEV void pv(const u32x4& v, const u32x4& pfrag) {                                                                                \
            asm volatile("v_mfma_f32_16x16x32_" SUF " a[%0:%1], %2, %3, a[%0:%1]" ::"n"(O0), "n"(O0 + 3), "v"(v), "v"(pfrag));                      \
        }                                                                                                                                         \
        template <int L0>                                                                                                                         \
        static FA_DEV void lsum(const u32x4& pfrag) {                                                                                              \
            asm volatile("v_mfma_f32_16x16x32_" SUF " a[%0:%1], a[208:211], %2, a[%0:%1]" ::"n"(L0), "n"(L0 + 3), "v"(pfrag));                      \
        }                                                                                                                                         \
    };
FA_W4_ASM(_Float16, "f16")
FA_W4_ASM(__bf16, "bf16")
#undef FA_W4_ASM

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(kW4Threads, 1) void fa_fwd_w4_kernel(const FwdKernelParams p) {
    constexpr int D = 128, KS = 4, DB = 8, ROWB = D * 2, SLOTS = D / 8, BN = 64, NKB = 4, NC = 2;
    constexpr int TILEB = BN * ROWB, RING = 3;
    constexpr int RINGB = 2 * RING * TILEB, STAGEB = kW4BlockM * ROWB;      // 96 KiB of rings + 64 KiB for O on its way out
    __shared__ __attribute__((aligned(16))) char smem_raw[RINGB + STAGEB];
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* kring = smem;
    FA_LDS char* vring = smem + RING * TILEB;
    FA_LDS char* stage = smem + RINGB;
    asm volatile("" ::: "a255");      // the kernel descriptor must allot all 256 accumulation registers

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi_g = (0x2130 >> (4 * g)) & 3;             // kPi  = {0, 3, 1, 2}   (fa_fwd_pp16.hip)
    const int pi2_g = (0x3120 >> (4 * g)) & 3;            // kPi2 = {0, 2, 1, 3}
    const int q_row_a = wave * 64 + n16;                  // this lane's query rows inside the 256-row block: q_row_a + 16 * qb
    const float c = p.scale_log2e;
    const uint32_t q_rowb = (uint32_t)(p.q.row * 2), k_rowb = (uint32_t)(p.k.row * 2), v_rowb = (uint32_t)(p.v.row * 2), o_rowb = (uint32_t)(p.o.row * 2);

    int tile, batch, head, tiles_seq;
    if (!decode_work<kW4BlockM>(blockIdx.x, p.n_q_tiles, 0u, nullptr, p.b, p.h, tile, batch, head, tiles_seq)) return;
    const int sq = p.seqlen_q, sk = p.seqlen_k;
    const int n_tiles = sk / BN;                          // (launcher: dense, no mask, whole tiles only)
    const int m0 = tile * kW4BlockM;
    const char* q_blk = (const char*)uniform_ptr((const T*)p.q_ptr + (int64_t)batch * p.q.batch + (int64_t)m0 * p.q.row + (int64_t)head * p.q.head);
    char* o_blk = (char*)uniform_ptr((T*)p.o_ptr + (int64_t)batch * p.o.batch + (int64_t)m0 * p.o.row + (int64_t)head * p.o.head);
    float* lse_blk = uniform_ptr(p.lse_ptr + ((int64_t)batch * p.h + head) * p.lse_row_stride + m0);
    const int head_k = head / p.h_ratio;
    const srd_t k_srd = make_srd(uniform_ptr((const T*)p.k_ptr + (int64_t)batch * p.k.batch + (int64_t)head_k * p.k.head), (uint32_t)(sk - 1) * k_rowb + ROWB);
    const srd_t v_srd = make_srd(uniform_ptr((const T*)p.v_ptr + (int64_t)batch * p.v.batch + (int64_t)head_k * p.v.head), (uint32_t)(sk - 1) * v_rowb + ROWB);
    (void)sq;

    // ---- lane constants ------------------------------------------------------------------------------------------
    constexpr int DPW = BN * SLOTS / kW4Threads;          // 4 LDS-DMA pieces (1 KiB) per wave and tile
    uint32_t dma_goff_k[DPW], dma_goff_v[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int chunk = (wave * DPW + i) * 64 + lane, row = chunk / SLOTS, phys = chunk % SLOTS;
        const int slot = lds_tile_logical_slot<D>(row, phys);
        dma_goff_k[i] = row * k_rowb + slot * 16;
        dma_goff_v[i] = row * v_rowb + slot * 16;
    }
    const uint32_t dma_loff = (uint32_t)wave * DPW * 1024;
    const uint32_t lds_k0 = lds_addr(kring) + dma_loff, lds_v0 = lds_addr(vring) + dma_loff;
    uint32_t k_rd[KS], v_rd[DB];
    {
        const int row = 4 * ((0x3120 >> (4 * (n16 >> 2))) & 3) + (n16 & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] = lds_addr(kring) + lds_tile_off<D>(row, 4 * ks + pi_g);
#pragma unroll
        for (int db = 0; db < DB; ++db) v_rd[db] = lds_addr(vring) + lds_tile_off<D>(4 * pi2_g + (n16 >> 2), 2 * db + ((n16 & 3) >> 1)) + 8 * (n16 & 1);
    }
    auto dma_k_piece = [&](int t, int slot, int i) __attribute__((always_inline)) {
        dma16_to_lds_hidden<false>(k_srd, (uint32_t)(t * BN) * k_rowb + dma_goff_k[i], lds_k0 + slot * TILEB + i * 1024);
    };
    auto dma_v_piece = [&](int t, int slot, int i) __attribute__((always_inline)) {
        dma16_to_lds_hidden<false>(v_srd, (uint32_t)(t * BN) * v_rowb + dma_goff_v[i], lds_v0 + slot * TILEB + i * 1024);
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < DPW; ++i) { dma_k_piece(0, 0, i); dma_v_piece(0, 0, i); }
#pragma unroll
    for (int i = 0; i < DPW; ++i) { dma_k_piece(1, 1, i); dma_v_piece(1, 1, i); }
#pragma unroll
    for (int i = 0; i < DPW; ++i) dma_k_piece(2, 2, i);
    {
        const rsrc_t q_rs = make_rsrc(q_blk, (uint32_t)(kW4BlockM - 1) * q_rowb + ROWB);
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            static_for<0, 4>([&](auto qbc) {
                constexpr int qb = decltype(qbc)::value;
                const u32x4 t = buf_load16(q_rs, (uint32_t)(q_row_a + 16 * qb) * q_rowb + (4 * ks + pi_g) * 16);
                acc_write<kAccQ + 16 * ks + 4 * qb + 0>(t.x);
                acc_write<kAccQ + 16 * ks + 4 * qb + 1>(t.y);
                acc_write<kAccQ + 16 * ks + 4 * qb + 2>(t.z);
                acc_write<kAccQ + 16 * ks + 4 * qb + 3>(t.w);
            });
        });
        static_for<0, 128>([&](auto rc) { acc_write<kAccO + decltype(rc)::value>(0u); });
        static_for<0, 16>([&](auto rc) { acc_write<kAccL + decltype(rc)::value>(0u); });
        static_for<0, 4>([&](auto rc) { acc_write<kAccOnes + decltype(rc)::value>(LP<T>::kOnes2); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- register-resident state ---------------------------------------------------------------------------------------
    f32x4 s01[NKB][2];                 // S^T of query columns 0, 1 (one buffer: written in A(u), read in B(u))
    f32x4 s23[2][NKB][2];              // S^T of query columns 2, 3, by tile parity (tile u's is read in A(u) while A(u) writes tile u+1's)
    u32x4 p01[2][NC][2];               // P^T of columns 0, 1, by tile parity (tile u's feeds B(u) while B(u) writes tile u+1's)
    u32x4 p23[NC][2];                  // P^T of columns 2, 3 (written in A(u), read in B(u))
    float m_run[4] = {kNegBig, kNegBig, kNegBig, kNegBig};
    int slot_u = 0, slot_u1 = 1, slot_u2 = 2;      // ring slots of tiles u, u+1, u+2 (K(u+3) goes where K(u) was)

    constexpr int PF = 2, NFB = PF + 1;            // LDS fragments in flight ahead of their MFMAs
    auto k_frag = [&](const uint32_t (&kb_addr)[KS], int f) __attribute__((always_inline)) -> u32x4 {      // f = 8 * (kb >> 1) + 2 * ks + (kb & 1)
        const int kbp = f >> 3, ks = (f >> 1) & 3, kb = 2 * kbp + (f & 1);
        return lds_read16((const FA_LDS char*)(uintptr_t)kb_addr[ks], 16 * kb * ROWB);
    };
    auto v_frag = [&](const uint32_t (&vb_addr)[DB], int f) __attribute__((always_inline)) -> u32x4 {      // f = 8 * chunk + db
        const int cch = f >> 3, db = f & 7;
        const u32x2 a0 = lds_read_tr8((const FA_LDS char*)(uintptr_t)vb_addr[db], (32 * cch) * ROWB);
        const u32x2 a1 = lds_read_tr8((const FA_LDS char*)(uintptr_t)vb_addr[db], (32 * cch + 16) * ROWB);
        return u32x4{a0.x, a0.y, a1.x, a1.y};
    };
    // QK^T MFMA number i of a tile (0..63): key-block pair outer, then k-step, key block, query column: every K fragment feeds four
    // consecutive MFMAs and an accumulator comes round again after 8 MFMAs
    auto qk_mfma = [&](auto ic, auto parc, const u32x4& kf) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, NP = decltype(parc)::value;      // NP = parity of the tile being computed
        constexpr int kbp = i / 32, ks = (i % 32) / 8, kb = 2 * kbp + (i % 8) / 4, qb = i % 4;
        f32x4& dst = qb < 2 ? s01[kb][qb] : s23[NP][kb][qb - 2];
        if constexpr (ks == 0) W4<T>::template qk_zero<kAccQ + 16 * ks + 4 * qb>(dst, kf);
        else W4<T>::template qk_acc<kAccQ + 16 * ks + 4 * qb>(dst, kf);
    };
    // P.V MFMA number i of a tile (0..71): per 32-key chunk 8 d blocks x 4 query columns, then the chunk's 4 row-sum MFMAs
    auto pv_mfma = [&](auto ic, auto parc, const u32x4& vf) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, PAR = decltype(parc)::value;
        constexpr int cch = i / 36, r = i % 36;
        if constexpr (r < 32) {
            constexpr int db = r / 4, qb = r % 4;
            W4<T>::template pv<kAccO + 16 * db + 4 * qb>(vf, qb < 2 ? p01[PAR][cch][qb] : p23[cch][qb - 2]);
        } else {
            constexpr int qb = r - 32;
            W4<T>::template lsum<kAccL + 4 * qb>(qb < 2 ? p01[PAR][cch][qb] : p23[cch][qb - 2]);
        }
    };

    // ---- softmax of two query columns as a list of single instructions (woven between MFMAs) -----------------------------------------
    // ops 0..1: -m * c of the two columns; then per (column, key block): 4 x fma, 4 x exp2, 2 x pack; then the guard (largest packed P of the
    // lane, see fa_fwd_pp16.hip): NOPS in all.  `S` / `P` are the arrays of the column pair, q0 its first query column.
    constexpr int NOPS = 2 + 80 + 11;
    float mc[2], xt[4];
    uint32_t gt[2];
    bool over = false;
    auto sm_op = [&](auto kc, f32x4 (&S)[NKB][2], u32x4 (&P)[NC][2], auto q0c) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, q0 = decltype(q0c)::value;
        if constexpr (k < 2) {
            mc[k] = -m_run[q0 + k] * c;
        } else if constexpr (k < 82) {
            constexpr int un = (k - 2) / 10, w = (k - 2) % 10, ql = un / 4, kb = un % 4;
            if constexpr (w < 4) xt[w] = __builtin_fmaf(S[kb][ql][w], c, mc[ql]);
            else if constexpr (w < 8) xt[w - 4] = fast_exp2(xt[w - 4]);
            else if constexpr (w == 8) { if constexpr (kb & 1) P[kb >> 1][ql].z = LP<T>::pack2(xt[0], xt[1]); else P[kb >> 1][ql].x = LP<T>::pack2(xt[0], xt[1]); }
            else { if constexpr (kb & 1) P[kb >> 1][ql].w = LP<T>::pack2(xt[2], xt[3]); else P[kb >> 1][ql].y = LP<T>::pack2(xt[2], xt[3]); }
        } else {
            constexpr int w = k - 82;
            if constexpr (w == 0) gt[0] = pk_max3_f16_bits(P[0][0].x, P[0][0].y, P[0][0].z);
            else if constexpr (w == 1) gt[1] = pk_max3_f16_bits(P[0][1].x, P[0][1].y, P[0][1].z);
            else if constexpr (w == 2) gt[0] = pk_max3_f16_bits(gt[0], P[0][0].w, P[1][0].x);
            else if constexpr (w == 3) gt[1] = pk_max3_f16_bits(gt[1], P[0][1].w, P[1][1].x);
            else if constexpr (w == 4) gt[0] = pk_max3_f16_bits(gt[0], P[1][0].y, P[1][0].z);
            else if constexpr (w == 5) gt[1] = pk_max3_f16_bits(gt[1], P[1][1].y, P[1][1].z);
            else if constexpr (w == 6) gt[0] = pk_max3_f16_bits(gt[0], P[1][0].w, gt[1]);
            else if constexpr (w == 7) gt[0] = pk_max3_f16_bits(gt[0], P[1][1].w, P[1][1].w);
            else if constexpr (w == 8) gt[1] = gt[0] << 16;
            else if constexpr (w == 9) gt[0] = max(gt[0], gt[1]);
            else over = gt[0] > ((LP<T>::kBits64 << 16) | 0xffffu);
        }
    };
    auto sm_ops = [&](auto lo, auto hi, f32x4 (&S)[NKB][2], u32x4 (&P)[NC][2], auto q0c) __attribute__((always_inline)) {
        static_for<decltype(lo)::value, decltype(hi)::value>([&](auto kc) { sm_op(kc, S, P, q0c); });
    };
    auto max4 = [&](float x) __attribute__((always_inline)) -> float { return max_four_groups(x); };
    // rare: some P of the column pair came out above 2^kW4DeferLog2 (or not finite).  Nothing of these columns is pending (see the header):
    // refresh their running max, rescale their O and l, run the pass again.
    auto refresh = [&](f32x4 (&S)[NKB][2], u32x4 (&P)[NC][2], auto q0c) __attribute__((always_inline)) {
        constexpr int q0 = decltype(q0c)::value;
        float mx[2];
#pragma unroll
        for (int ql = 0; ql < 2; ++ql) {
            mx[ql] = S[0][ql][0];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx[ql] = fmaxf(mx[ql], S[kb][ql][r]);
            mx[ql] = max4(mx[ql]);
        }
        if (__builtin_amdgcn_ballot_w64((mx[0] - m_run[q0]) * c > kW4DeferLog2 || (mx[1] - m_run[q0 + 1]) * c > kW4DeferLog2) == 0) return;
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");      // every MFMA that wrote these accumulators has retired
        static_for<0, 2>([&](auto qlc) {
            constexpr int ql = decltype(qlc)::value, qb = q0 + ql;
            const float m_new = fmaxf(m_run[qb], mx[ql]);
            const float alpha = fast_exp2((m_run[qb] - m_new) * c);
            m_run[qb] = m_new;
            static_for<0, DB>([&](auto dbc) {
                static_for<0, 4>([&](auto rc) { acc_scale<kAccO + 16 * decltype(dbc)::value + 4 * qb + decltype(rc)::value>(alpha); });
            });
            static_for<0, 4>([&](auto rc) { acc_scale<kAccL + 4 * qb + decltype(rc)::value>(alpha); });
        });
        sm_ops(std::integral_constant<int, 0>{}, std::integral_constant<int, 82>{}, S, P, q0c);
        asm volatile("s_nop 4" ::: "memory");
    };
    auto check = [&](f32x4 (&S)[NKB][2], u32x4 (&P)[NC][2], auto q0c) __attribute__((always_inline)) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(over) != 0, 0)) refresh(S, P, q0c);
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;

    // ---- phases ---------------------------------------------------------------------------------------------------------
    // A: S(t)^T = K(t) Q^T for the tile t in ring slot `slot_k`, parity NP; softmax ops of columns 2, 3 of tile t - 1 woven in when WEAVE
    auto phase_a = [&](int slot_k, auto npc, auto weave, int dma_t, int dma_slot) __attribute__((always_inline)) {
        constexpr int NP = decltype(npc)::value;
        constexpr bool WEAVE = decltype(weave)::value;
        uint32_t kb_addr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kb_addr[ks] = k_rd[ks] + slot_k * TILEB; asm volatile("" : "+v"(kb_addr[ks])); }
        u32x4 fr[NFB];
        static_for<0, PF>([&](auto fc) { fr[decltype(fc)::value] = k_frag(kb_addr, decltype(fc)::value); });
        static_for<0, 64>([&](auto ic) {
            constexpr int i = decltype(ic)::value, f = i / 4;
            if constexpr (i % 4 == 0 && f + PF < 16) fr[(f + PF) % NFB] = k_frag(kb_addr, f + PF);
            __builtin_amdgcn_sched_barrier(0);
            qk_mfma(ic, npc, fr[f % NFB]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WEAVE) {
                sm_ops(std::integral_constant<int, (i * NOPS) / 64>{}, std::integral_constant<int, ((i + 1) * NOPS) / 64>{}, s23[NP ^ 1], p23, c2{});
                if constexpr (i % 16 == 6) dma_k_piece(dma_t, dma_slot, i / 16);      // K(u+3), one piece every 16 MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    // B: O^T += V(t)^T P(t)^T for the tile t in ring slot `slot_v`, parity PAR; softmax ops of columns 0, 1 of tile t + 1 woven in when WEAVE
    auto phase_b = [&](int slot_v, auto parc, auto weave, int dma_t, int dma_slot) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        constexpr bool WEAVE = decltype(weave)::value;
        uint32_t vb_addr[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) { vb_addr[db] = v_rd[db] + slot_v * TILEB; asm volatile("" : "+v"(vb_addr[db])); }
        u32x4 fr[NFB];
        static_for<0, PF>([&](auto fc) { fr[decltype(fc)::value] = v_frag(vb_addr, decltype(fc)::value); });
        asm volatile("s_nop 1" ::: "memory");      // (P written by VALU just before: MFMA source hazard)
        static_for<0, 72>([&](auto ic) {
            constexpr int i = decltype(ic)::value, cch = i / 36, r = i % 36, f = cch * 8 + (r < 32 ? r / 4 : 7);
            if constexpr (r < 32 && r % 4 == 0 && f + PF < 16) fr[(f + PF) % NFB] = v_frag(vb_addr, f + PF);
            __builtin_amdgcn_sched_barrier(0);
            pv_mfma(ic, parc, fr[f % NFB]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WEAVE) {
                sm_ops(std::integral_constant<int, (i * NOPS) / 72>{}, std::integral_constant<int, ((i + 1) * NOPS) / 72>{}, s01, p01[PAR ^ 1], c0{});
                if constexpr (i % 18 == 7) dma_v_piece(dma_t, dma_slot, i / 18);      // V(u+2)
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // ---- tile 0: scores, running max seeded with the tile's row maxima, P of columns 0, 1 ------------------------------------------------
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    phase_a(0, c0{}, no{}, 0, 0);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // (asm-issued MFMAs: the scores must have landed before the VALU reads them)
    static_for<0, 4>([&](auto qbc) {
        constexpr int qb = decltype(qbc)::value;
        float mx = qb < 2 ? s01[0][qb][0] : s23[0][0][qb - 2][0];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, qb < 2 ? s01[kb][qb][r] : s23[0][kb][qb - 2][r]);
        m_run[qb] = fmaxf(kNegBig, max4(mx));
    });
    sm_ops(c0{}, std::integral_constant<int, NOPS>{}, s01, p01[0], c0{});
    check(s01, p01[0], c0{});

    // ---- steady state: step(u) for u = 0 .. n_tiles - 2, two per trip (tile parity is a compile-time constant) ---------------------------
    auto advance = [&]() __attribute__((always_inline)) {
        const int t = slot_u;
        slot_u = slot_u1; slot_u1 = slot_u2; slot_u2 = t;
    };
    auto step = [&](int u, auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        using npar = std::integral_constant<int, PAR ^ 1>;
        // K(u+1), V(u) have landed (the 8 pieces of step u-1 may still fly); every wave is done with K(u) and V(u-1), whose slots are refilled below
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        phase_a(slot_u1, npar{}, yes{}, u + 3, slot_u);           // S(u+1); softmax of S(u) columns 2, 3; K(u+3) -> where K(u) was
        check(s23[PAR], p23, c2{});
        phase_b(slot_u, parc, yes{}, u + 2, slot_u2);             // O += V(u) P(u); softmax of S(u+1) columns 0, 1; V(u+2) -> where V(u-1) was
        check(s01, p01[PAR ^ 1], c0{});
        advance();
    };
    int u = 0;
    for (; u + 2 <= n_tiles - 1; u += 2) {
        step(u, c0{});
        step(u + 1, c1{});
    }
    // ---- last tile: its columns 2, 3 (nothing left to hide them behind), its P.V ---------------------------------------------------------
    auto last = [&](auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        sm_ops(c0{}, std::integral_constant<int, NOPS>{}, s23[PAR], p23, c2{});
        check(s23[PAR], p23, c2{});
        phase_b(slot_u, parc, no{}, 0, 0);
    };
    if (u < n_tiles - 1) {
        step(u, c0{});
        last(c1{});
    } else {
        last(c0{});
    }

    // ---- epilogue (wave-local): normalise, round, stage the wave's 64 rows in LDS, store whole rows ---------------------------------------------
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    static_for<0, 4>([&](auto qbc) {
        constexpr int qb = decltype(qbc)::value;
        const float l_tot = acc_read<kAccL + 4 * qb>();
        const float inv = l_tot != 0.f ? fast_rcp(l_tot) : 0.f;
        const float lse = l_tot != 0.f ? (m_run[qb] * c + fast_log2(l_tot)) * kLn2 : 0.f;
        const int row = q_row_a + 16 * qb;
        if (g == 0) lse_blk[row] = lse;
        static_for<0, DB>([&](auto dbc) {
            constexpr int db = decltype(dbc)::value, o0 = kAccO + 16 * db + 4 * qb;
            u32x2 w;
            w.x = LP<T>::pack2(acc_read<o0 + 0>() * inv, acc_read<o0 + 1>() * inv);
            w.y = LP<T>::pack2(acc_read<o0 + 2>() * inv, acc_read<o0 + 3>() * inv);
            lds_write8(stage, lds_tile_off<D>(row, 2 * db + (g >> 1)) + 8 * (g & 1), w);      // d = 16*db + 4*g + {0..3}
        });
    });
    const rsrc_t o_rs = make_rsrc(o_blk, (uint32_t)(kW4BlockM - 1) * o_rowb + ROWB);
    constexpr int O_CHUNKS = (64 * SLOTS) / 64;
#pragma unroll
    for (int i = 0; i < O_CHUNKS; ++i) {
        const int chunk = lane + i * 64, row = wave * 64 + chunk / SLOTS, slot = chunk % SLOTS;
        buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, lds_read16(stage, lds_tile_off<D>(row, slot)));
    }
}

// eligibility (this file's first form): dense, no mask, whole 256-row / 64-key tiles
bool fwd_w4_eligible(const FwdKernelParams& kp) {
    return kp.d == 128 && !kp.is_causal && kp.cu_seqlens_q == nullptr && kp.seqlen_q % kW4BlockM == 0 && kp.seqlen_k % 64 == 0 && kp.seqlen_k >= 64;
}

hipError_t launch_fwd_w4(const FwdKernelParams& kp, int dtype, uint32_t grid, hipStream_t stream) {
    if (grid == 0) return hipSuccess;
    if (dtype == 0) hipLaunchKernelGGL((fa_fwd_w4_kernel<_Float16, false>), dim3(grid), dim3(kW4Threads), 0, stream, kp);
    else hipLaunchKernelGGL((fa_fwd_w4_kernel<__bf16, false>), dim3(grid), dim3(kW4Threads), 0, stream, kp);
    return hipGetLastError();
}

}  // namespace fa
