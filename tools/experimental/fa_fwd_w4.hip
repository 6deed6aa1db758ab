// ==== EXPERIMENT, NOT PART OF THE PRODUCT (round 2).  Not built by build.py, not dispatched, kept for the record. ====
// Status when parked: correct on fp16 and on non-causal bf16, wrong on causal bf16 (hazards of the kind described below that were not all
// found); 930 TFLOP/s at b4 s8192 h32 d128 against 1186 for the ping-pong kernel in the same process (gpurun_out r2 t18): the simple
// two-phase schedule below leaves phase B VALU-bound (290 VALU in 32 MFMA gaps).  A balanced schedule needs the softmax of tile u to
// overlap QK^T of tile u+1 as well, i.e. a second S buffer (64 more VGPRs than the 213 used here).  What the experiment DID establish
// about inline-asm MFMAs under hipcc (each found on hardware, each cost hours):
//   1. any C++ arithmetic on an "+a" accumulator inside the tile loop, or any branch around asm statements that use one, makes hipcc
//      shuttle all 128 accumulator registers through VGPRs on every iteration;
//   2. hipcc re-homes accumulator tiles (v_accvgpr_write / _mov / _read) at region boundaries - loop exit, branch joins - right next to
//      asm MFMAs whose hazards it cannot see: XDL-write -> accvgpr_read needs 11+ wait states, accvgpr_write -> MFMA SrcC needs 2; the
//      symptom is stale data in exactly the registers the copy touched last (register 0 of two tiles here);
//   3. "+a" / "a" operands must be re-defined in the accumulator class once (asm volatile("" : "+a"(x))) or hipcc copies four registers
//      into an AGPR temporary in front of every MFMA.
// To build it anyway: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -I flash-attention-turing_amd/csrc -I include -c
// fa_fwd_w4.hip — fused attention forward for MI355X (gfx950), head_dim 128: one wave per SIMD, 64 query rows per wave.
//
// Same contract, tile image, K / V rings and fragment layouts as fa_fwd_pp.hip (which keeps head_dim 64); what changes is who
// shares what.  The ping-pong kernel puts two 32-row waves on every SIMD; measured there (tools/phase_timing.py, timing-only VALU
// ablations, profiles/r2_fwd_*): the SIMD's ISSUE slots are the limit - 2 x (167 VALU + 48 LDS reads) per 64 MFMAs = 6.7 non-MFMA
// instructions per MFMA against ~5 that fit under a 32-cycle MFMA - and the period of the two groups is the SUM of their matrix
// phases.  Here a workgroup is 4 waves (256 threads, one per SIMD, the whole 512-register file each) and a wave owns TWO 32-row
// query blocks:
//   * every K fragment (ds_read_b128) and every V^T fragment (2 x ds_read_b64_tr_b16) feeds two MFMAs - 48 LDS reads per 64
//     MFMAs instead of 96;
//   * O^T (128 registers) and Q (64) live in the accumulator half of the register file and are only ever touched by MFMAs
//     (inline asm "+a" / "a" operands), the softmax works in the architectural VGPRs;
//   * per 64-key tile a wave runs two phases: A = S(u) = K(u) Q^T (32 MFMAs, fillers: the fragment reads and the LDS-DMA of the
//     tiles two / one ahead), B = O += V(u-1)^T P(u-1) (32 MFMAs) with the whole softmax of tile u as fillers, in fixed
//     per-MFMA slices pinned with sched_barrier (hipcc has no latency model for asm MFMAs: source order IS the schedule);
//   * one workgroup barrier per tile;
//   * the FAST path never rescales O: the running max of a row is fixed by the first tile in which the row sees a key (O and l are
//     still 0 then, so nothing needs scaling) and later tiles are exponentiated against it.  A row whose max then grows by more
//     than 2^kW4DeferLog2 raises a flag; the workgroup finishes the block anyway, discards it and redoes it with the rescaling
//     one-q-block-at-a-time loop at the end of the kernel (exact online softmax, ~2x slower, data-dependent and rare: never on
//     N(0,1) inputs, forced in tests/test_attention_gpu.py::test_online_softmax_rescale_spike).  Reason: O lives in AGPRs as
//     inline-asm operands; any C++ arithmetic on it inside the tile loop makes hipcc shuttle all 128 accumulator registers through
//     VGPRs on EVERY tile (960 v_accvgpr moves per iteration in the first version of this kernel).
// MFMAs are issued from inline asm, so hipcc's hazard recogniser does not see them; the schedule keeps every producer /
// consumer pair apart by construction (comments at the sites) and pads the few places where it cannot.  One hazard is NOT
// about results: an MFMA issued right behind another one waits in the matrix pipe's queue and reads its A / B registers only
// when it starts (up to 32 cycles later), while hipcc believes an asm statement has consumed its inputs the moment it is issued
// and happily lets the next VALU instruction reuse those registers (first version of this kernel: P*V wrong in exactly the
// tiles whose P fragment was overwritten by v_cvt_pk five instructions after the MFMA).  `hold()` below keeps the operands of
// every MFMA alive until two further MFMAs have been issued.
#include "fa_device.hpp"
#include "fa_params.hpp"

#include <type_traits>

namespace fa {

// hand-scheduled asm MFMA forms (see item 2 above for why one of them is padded)
template <typename T> struct W4Asm;
template <> struct W4Asm<_Float16> {
    // no hazard pad inside - the CALLER guarantees that a / b were not
    // written by a VALU instruction in the two preceding issue slots and that nothing but an MFMA chained through `acc` touches `acc`
    // for 12 slots after it.  *_aq: B operand (Q fragment) lives in the accumulator half of the register file.
    // The VGPR source operands are declared READ-WRITE ("+v") although the instruction only reads them: an MFMA issued behind
    // another one reads its A / B registers up to a full MFMA later, while hipcc regards an asm statement's inputs as consumed at
    // issue and may keep the value in a second register and hand the first one to the very next VALU instruction (seen on
    // hardware).  A read-write operand has exactly one home; the caller keeps it alive two MFMAs longer with hold().
    static FA_DEV void mfma_o(f32x16& acc, u32x4& a, u32x4& b) {               // O^T (AGPR) += a * b
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc), "+v"(a), "+v"(b));
    }
    // padded form for code that hipcc may have prefixed with register copies it made up itself (region boundaries: it re-homes
    // accumulator tiles with v_accvgpr_write right in front of the statement, and a VALU write needs wait states before an MFMA
    // reads the register - found on hardware: register 0 of two tiles, the last one written by the copy, was read stale)
    static FA_DEV void mfma_o_padded(f32x16& acc, u32x4& a, u32x4& b) {
        asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc), "+v"(a), "+v"(b));
    }
    static FA_DEV void mfma_s(f32x16& acc, u32x4& a, const u32x4& b_in_agpr) { // S^T (VGPR) += a * b
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc), "+v"(a) : "a"(b_in_agpr));
    }
    static FA_DEV void mfma_s0(f32x16& acc, u32x4& a, const u32x4& b_in_agpr) {// S^T (VGPR) = a * b  (C = 0)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc), "+v"(a) : "a"(b_in_agpr));
    }
};
template <> struct W4Asm<__bf16> {
    // no hazard pad inside - the CALLER guarantees that a / b were not
    // written by a VALU instruction in the two preceding issue slots and that nothing but an MFMA chained through `acc` touches `acc`
    // for 12 slots after it.  *_aq: B operand (Q fragment) lives in the accumulator half of the register file.
    // The VGPR source operands are declared READ-WRITE ("+v") although the instruction only reads them: an MFMA issued behind
    // another one reads its A / B registers up to a full MFMA later, while hipcc regards an asm statement's inputs as consumed at
    // issue and may keep the value in a second register and hand the first one to the very next VALU instruction (seen on
    // hardware).  A read-write operand has exactly one home; the caller keeps it alive two MFMAs longer with hold().
    static FA_DEV void mfma_o(f32x16& acc, u32x4& a, u32x4& b) {               // O^T (AGPR) += a * b
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc), "+v"(a), "+v"(b));
    }
    // padded form for code that hipcc may have prefixed with register copies it made up itself (region boundaries: it re-homes
    // accumulator tiles with v_accvgpr_write right in front of the statement, and a VALU write needs wait states before an MFMA
    // reads the register - found on hardware: register 0 of two tiles, the last one written by the copy, was read stale)
    static FA_DEV void mfma_o_padded(f32x16& acc, u32x4& a, u32x4& b) {
        asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc), "+v"(a), "+v"(b));
    }
    static FA_DEV void mfma_s(f32x16& acc, u32x4& a, const u32x4& b_in_agpr) { // S^T (VGPR) += a * b
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc), "+v"(a) : "a"(b_in_agpr));
    }
    static FA_DEV void mfma_s0(f32x16& acc, u32x4& a, const u32x4& b_in_agpr) {// S^T (VGPR) = a * b  (C = 0)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc), "+v"(a) : "a"(b_in_agpr));
    }
};

constexpr int kW4Threads = 256;
constexpr int kW4BlockM = 256;
constexpr int kW4BlockN = 64;
constexpr float kW4DeferLog2 = 6.0f;

// keep a register value alive (unmodified, not reusable) up to this point of the instruction stream
FA_DEV void hold(u32x4& x) { asm volatile("" : "+v"(x)); }

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(kW4Threads, 1) void fa_fwd_w4_kernel(const FwdKernelParams p) {
    constexpr int D = 128, KS = D / 16, DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    constexpr int TILEB = kW4BlockN * ROWB;
    constexpr int RING = 3;
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * RING * TILEB];      // 96 KiB; the O block (64 KiB) aliases it in the epilogue
    FA_LDS char* smem = (FA_LDS char*)smem_raw;
    FA_LDS char* kring = smem;
    FA_LDS char* vring = smem + RING * TILEB;

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tile, batch, head, tiles_seq;
    if (!decode_work<kW4BlockM>(blockIdx.x, p.n_q_tiles, p.varlen_slots, p.cu_seqlens_q, p.b, p.h, tile, batch, head, tiles_seq)) return;
    if (CAUSAL) tile = tiles_seq - 1 - tile;      // heaviest (latest) query tiles first
    const int head_k = head / p.h_ratio;

    int sq = p.seqlen_q, sk = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int64_t q_boff = (int64_t)batch * p.q.batch, k_boff = (int64_t)batch * p.k.batch,
            v_boff = (int64_t)batch * p.v.batch, o_boff = (int64_t)batch * p.o.batch;
    if (p.cu_seqlens_q != nullptr) {
        const int q_beg = p.cu_seqlens_q[batch], k_beg = p.cu_seqlens_k[batch];
        sq = min(p.cu_seqlens_q[batch + 1] - q_beg, p.seqlen_q);      // clamp to the declared max_seqlen_q (padded LSE rows)
        sk = p.cu_seqlens_k[batch + 1] - k_beg;
        q_row0 = q_beg; k_row0 = k_beg;
        q_boff = k_boff = v_boff = o_boff = 0;
    }
    const int m0 = tile * kW4BlockM;
    if (m0 >= sq) return;
    const int delta = sk - sq;
    const int rows_here = min(kW4BlockM, sq - m0);

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

    int n_tiles = (sk + kW4BlockN - 1) / kW4BlockN;
    if (CAUSAL) {
        const int max_key = m0 + rows_here - 1 + delta;
        n_tiles = max_key < 0 ? 0 : min(n_tiles, max_key / kW4BlockN + 1);
    }
    const float c = p.scale_log2e;

    // ---- lane constants --------------------------------------------------------------------------------
    // wave w owns query rows [64w, 64w + 64) of the block: q-block 0 = rows 64w + l31, q-block 1 = rows 64w + 32 + l31
    const int wave_q_lo = m0 + wave * 64;
    constexpr int DPW = SLOTS / 4;            // 1-KiB LDS-DMA pieces per wave per tile (4 waves x 4 = the 16 KiB tile)
    uint32_t dma_goff_k[DPW], dma_goff_v[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int chunk = (wave * DPW + i) * 64 + lane;
        const int row = chunk / SLOTS, phys = chunk % SLOTS;
        const int slot = lds_tile_logical_slot<D>(row, phys);
        dma_goff_k[i] = row * k_rowb + slot * 16;
        dma_goff_v[i] = row * v_rowb + slot * 16;
    }
    const uint32_t dma_loff = (uint32_t)wave * DPW * 1024;
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

    // Q fragments of both q-blocks: B operands for the whole kernel, parked in AGPRs by the "a" constraints of the QK^T MFMAs
    u32x4 qf[2][KS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qb][ks] = buf_load16(q_rs, (uint32_t)(wave * 64 + qb * 32 + l31) * q_rowb + (2 * ks + hi) * 16);
    // re-define every fragment in the accumulator register class once: otherwise hipcc keeps Q in VGPRs (where the loads put it)
    // and copies 4 registers into an AGPR temporary in front of every QK^T MFMA
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(qf[qb][ks]));

    f32x16 oacc[DB][2];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[db][qb][r] = 0.f;
    float m_run[2] = {kNegBig, kNegBig}, l_run[2] = {0.f, 0.f};

    auto dma_k_piece = [&](int t, int slot, int i) {
        dma16_to_lds_hidden<false>(k_srd, (uint32_t)(t * kW4BlockN) * k_rowb + dma_goff_k[i], lds_k0 + slot * TILEB + i * 1024);
    };
    auto dma_v_piece = [&](int t, int slot, int i) {
        dma16_to_lds_hidden<false>(v_srd, (uint32_t)(t * kW4BlockN) * v_rowb + dma_goff_v[i], lds_v0 + slot * TILEB + i * 1024);
    };

    // ---- prologue: K(0), V(0), K(1) ----
    if (n_tiles > 0) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) { dma_k_piece(0, 0, i); dma_v_piece(0, 0, i); dma_k_piece(1, 1, i); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (also the Q loads: they are parked in AGPRs long before their first MFMA)
    __syncthreads();

    f32x16 sacc[2][2];                                // [key block][q-block]: S^T tiles of the current key tile
    u32x4 pcur[4][2], pnext[4][2];                    // [k-slice ts][q-block]: P(u-1) being multiplied, P(u) being produced
    int ring_u = 0, ring_um1 = 2, ring_up1 = 1;
    // operands of the last MFMAs of a phase, kept alive into the next phase (see `hold`)
    u32x4 keep_k[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, keep_v[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}},
          keep_p[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};

    // ---- phase A: S(u) = K(u) Q^T; fillers: fragment reads two k-steps ahead, the DMA of K(u+2) / V(u+1) ----
    auto phase_a = [&](int u) {
        // per-tile fragment addresses, pinned: one v_add per k-step (the key-block offset folds into the ds_read immediate);
        // left to itself hipcc re-associates the ring-slot offset into a scalar and spends one v_add per READ (57 per tile)
        uint32_t kaddr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kaddr[ks] = k_rd[ks] + (uint32_t)(ring_u * TILEB); asm volatile("" : "+v"(kaddr[ks])); }
        u32x4 kf[KS][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) kf[ks][kb] = lds_read16(kring, kaddr[ks] + kb * 32 * ROWB);
        const bool more_k = u + 2 < n_tiles, more_v = u + 1 < n_tiles;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 2 < KS) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) kf[ks + 2][kb] = lds_read16(kring, kaddr[ks + 2] + kb * 32 * ROWB);
            }
            // one LDS-DMA piece per k-step: K(u+2) into the slot K(u-1) left at the last barrier, V(u+1) into V(u-2)'s
            if (ks < DPW) { if (more_k) dma_k_piece(u + 2, ring_um1, ks); }
            else if (ks < 2 * DPW) { if (more_v) dma_v_piece(u + 1, ring_up1, ks - DPW); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (ks == 0) W4Asm<T>::mfma_s0(sacc[kb][qb], kf[ks][kb], qf[qb][ks]);
                    else W4Asm<T>::mfma_s(sacc[kb][qb], kf[ks][kb], qf[qb][ks]);
                }
            if (ks == 0) { hold(keep_v[0]); hold(keep_v[1]); hold(keep_p[0]); hold(keep_p[1]); }     // last P*V MFMAs of the previous tile have started
            if (ks >= 1) { hold(kf[ks - 1][0]); hold(kf[ks - 1][1]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        keep_k[0] = kf[KS - 1][0]; keep_k[1] = kf[KS - 1][1];
    };

    // ---- softmax of tile u, cut into 32 slices that ride in the MFMA gaps of phase B ----
    // state shared by the slices
    float mx[2], mcur[2], psum[2];
    bool grew = false;                                 // per lane: some tile outgrew this row's reference max -> redo the block
    auto sm_slice = [&](int j) {
        if (j < 8) {                                   // running max of the tile: S tile (kb, qb) = j / 2, registers [8h, 8h + 8)
            const int t = j >> 1, kb = t & 1, qb = t >> 1, h = j & 1;
            float m = (t & 1) == 0 && h == 0 ? sacc[kb][qb][0] : fmaxf(mx[qb], sacc[kb][qb][8 * h]);
#pragma unroll
            for (int r = 1; r < 8; ++r) m = fmaxf(m, sacc[kb][qb][8 * h + r]);
            mx[qb] = m;
        } else if (j == 8) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                mx[qb] = max_both_halves(mx[qb]);
                // the row's reference max is set ONCE, by the first tile in which it sees a key (O = l = 0 until then); growth by more
                // than 2^kW4DeferLog2 afterwards cannot be absorbed here (P would leave the exactly representable range): flag it
                const bool unseen = m_run[qb] == kNegBig;
                grew = grew || (!unseen && (mx[qb] - m_run[qb]) * c > kW4DeferLog2);
                m_run[qb] = (unseen && mx[qb] > -INFINITY) ? mx[qb] : m_run[qb];
                mcur[qb] = m_run[qb] * c;
                psum[qb] = 0.f;
            }
        } else if (j >= 10 && j < 26) {                // exp / row sum: 4 score elements of S tile (j - 10) / 4
            const int e = j - 10, t = e >> 2, kb = t & 1, qb = t >> 1, q4 = e & 3;
#pragma unroll
            for (int r = 4 * q4; r < 4 * q4 + 4; ++r) {
                const float pv = fast_exp2(__builtin_fmaf(sacc[kb][qb][r], c, -mcur[qb]));
                psum[qb] += pv;
                sacc[kb][qb][r] = pv;
            }
        } else if (j >= 26 && j < 30) {                // round P to the input type: B operands of the next phase B
            const int t = j - 26, kb = t & 1, qb = t >> 1;
            pnext[2 * kb][qb] = pack_c_half<T>(sacc[kb][qb], 0);
            pnext[2 * kb + 1][qb] = pack_c_half<T>(sacc[kb][qb], 1);
        }
    };
    auto sm_mask = [&](int u) {                        // diagonal / ragged tiles only: -inf outside the visible keys
        const int n0 = u * kW4BlockN;
        const bool need_mask = (n0 + kW4BlockN > sk) || (CAUSAL && (n0 + kW4BlockN - 1 > wave_q_lo + delta));
        if (need_mask) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int lim = CAUSAL ? min(sk - 1, wave_q_lo + qb * 32 + l31 + delta) : sk - 1;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = n0 + 32 * kb + c_row(r, hi);
                        sacc[kb][qb][r] = key <= lim ? sacc[kb][qb][r] : -INFINITY;
                    }
            }
        }
    };

    // ---- phase B: O += V(u-1)^T P(u-1) (32 MFMAs) with softmax(u) in the gaps ----
    auto phase_b = [&](auto have_pv_t, auto have_sm_t) {
        constexpr bool have_pv = decltype(have_pv_t)::value, have_sm = decltype(have_sm_t)::value;
        uint32_t vaddr[2][DB];
        if constexpr (have_pv) {
#pragma unroll
            for (int sec = 0; sec < 2; ++sec)
#pragma unroll
                for (int db = 0; db < DB; ++db) { vaddr[sec][db] = v_rd[sec][db] + (uint32_t)(ring_um1 * TILEB); asm volatile("" : "+v"(vaddr[sec][db])); }
        }
        if constexpr (!have_pv) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // first tile only: let the queued QK^T MFMAs start before anything reuses their K fragments
            hold(keep_k[0]); hold(keep_k[1]);
        }
        u32x4 vf[16];
        auto rd_v = [&](int f) {                       // fragment f: k-slice ts = f / 4, d-block db = f % 4
            const int ts = f >> 2, db = f & 3;
            const u32x2 a0 = lds_read_tr8(vring, vaddr[0][db] + ts * 16 * ROWB);
            const u32x2 a1 = lds_read_tr8(vring, vaddr[1][db] + ts * 16 * ROWB);
            return u32x4{a0.x, a0.y, a1.x, a1.y};
        };
        if constexpr (have_pv) { vf[0] = rd_v(0); vf[1] = rd_v(1); }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int f = j >> 1, qb = j & 1, ts = f >> 2, db = f & 3;
            if constexpr (have_pv) {
                if (qb == 0 && f + 2 < 16) vf[f + 2] = rd_v(f + 2);
                // Padded form (s_nop in front) wherever hipcc may have put register copies of its own right before the statement:
                // the FIRST MFMA on each accumulator tile in a phase (a phase starts a new basic block after the mask branch, and
                // block boundaries are where hipcc re-homes tiles with v_accvgpr_write / _mov), and the whole drain (loop exit).
                // Found on hardware twice: stale accumulator registers in exactly the tiles whose copies sat in front of an MFMA.
                if (have_sm && j >= 8) W4Asm<T>::mfma_o(oacc[db][qb], vf[f], pcur[ts][qb]);
                else W4Asm<T>::mfma_o_padded(oacc[db][qb], vf[f], pcur[ts][qb]);
                if (j == 1) {                                                      // the last MFMAs of the previous phase have started
                    hold(keep_k[0]); hold(keep_k[1]);
                    if constexpr (!have_sm) { hold(keep_v[0]); hold(keep_v[1]); hold(keep_p[0]); hold(keep_p[1]); }
                }
                if (j >= 3 && (j & 1)) hold(vf[(j - 3) >> 1]);                     // fragment f was last read by MFMA 2f + 1
                if (j >= 9 && (j & 7) < 2) hold(pcur[(j - 9) >> 3][j & 1]);        // P slice ts was last read by MFMAs 8ts + 6, 8ts + 7
            }
            if constexpr (have_sm) sm_slice(j);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (have_pv) { keep_v[0] = vf[14]; keep_v[1] = vf[15]; keep_p[0] = pcur[3][0]; keep_p[1] = pcur[3][1]; }
        if constexpr (have_sm) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) l_run[qb] += psum[qb];
        }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    // ---- main loop: one barrier per key tile.  No control flow around the P*V MFMAs anywhere (the first tile is peeled, blocks
    // without a single visible key tile leave early): a branch around asm statements whose operands are accumulator tiles makes
    // hipcc copy the tiles at the join ----
    auto tile_iteration = [&](int u, auto have_pv_t) {
        phase_a(u);
        const int n0 = u * kW4BlockN;
        const bool maybe_masked = (n0 + kW4BlockN > sk) || (CAUSAL && (n0 + kW4BlockN - 1 > m0 + delta));
        if (maybe_masked) {
            asm volatile("s_nop 15" ::: "memory");     // S was written by asm MFMAs a few slots ago
            sm_mask(u);
        }
        phase_b(have_pv_t, yes{});
#pragma unroll
        for (int ts = 0; ts < 4; ++ts)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) pcur[ts][qb] = pnext[ts][qb];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ring_um1 = ring_u; ring_u = ring_up1; ring_up1 = ring_up1 == 2 ? 0 : ring_up1 + 1;
        // Last tile: hipcc allocates the accumulator tiles differently after the loop and copies them (v_accvgpr_read / _mov) on the
        // exit edge - right behind P*V MFMAs it cannot see, i.e. inside their 11-wait-state (plus queue) shadow.  Found on
        // hardware: exactly the two tiles written by the last two MFMAs of the loop came out stale.  Pad INSIDE the loop body, so
        // that the pad precedes whatever sits on the edge.
        if (u + 1 == n_tiles) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    };
    if (n_tiles == 0) {
        // no key is visible to any row of the block (sk = 0, or causal with sk << sq): O = 0, LSE = 0 (reference dead-row convention)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int q_row = wave * 64 + qb * 32 + l31;
            if (hi == 0 && q_row < rows_here) lse_base[q_row] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < (64 * SLOTS) / 64; ++i) {
            const int chunk = lane + i * 64, row = wave * 64 + chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(o_rs, (uint32_t)row * o_rowb + slot * 16, u32x4{0u, 0u, 0u, 0u});
        }
        return;
    }
    tile_iteration(0, no{});
    for (int u = 1; u < n_tiles; ++u) tile_iteration(u, yes{});
    phase_b(yes{}, no{});                              // drain: P*V of the last tile
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    hold(keep_v[0]); hold(keep_v[1]); hold(keep_p[0]); hold(keep_p[1]);

    // ---- did any row of the block outgrow its reference max?  (workgroup-uniform decision through one LDS word) ----
    __syncthreads();                                   // every wave is done reading the rings
    FA_LDS uint32_t* flag = (FA_LDS uint32_t*)(smem + 2 * RING * TILEB - 16);
    if (tid == 0) *flag = 0u;
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(grew) != 0 && lane == 0) *flag = 1u;
    __syncthreads();
    const bool redo = __builtin_amdgcn_readfirstlane((int)*flag) != 0;

    // normalise, round, stage 32 rows per wave in LDS at `stage`, store them as whole rows (wave-local: no barrier)
    auto store_rows = [&](f32x16 (&acc)[DB], float m, float l, int row0_in_block, FA_LDS char* stage) {
        const int q_row = row0_in_block + l31;
        const float l_tot = sum_both_halves(l);
        const float inv = l_tot > 0.f ? fast_rcp(l_tot) : 0.f;
        const float lse = l_tot > 0.f ? (m * c + fast_log2(l_tot)) * kLn2 : 0.f;
        if (hi == 0 && q_row < rows_here) lse_base[q_row] = lse;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w.x = LP<T>::pack2(acc[db][4 * g4 + 0] * inv, acc[db][4 * g4 + 1] * inv);
                w.y = LP<T>::pack2(acc[db][4 * g4 + 2] * inv, acc[db][4 * g4 + 3] * inv);
                lds_write8(stage, lds_tile_off<D>(l31, 4 * db + g4) + 8 * hi, w);
            }
        constexpr int O_CHUNKS = (32 * SLOTS) / 64;
#pragma unroll
        for (int i = 0; i < O_CHUNKS; ++i) {
            const int chunk = lane + i * 64, row = chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(o_rs, (uint32_t)(row0_in_block + row) * o_rowb + slot * 16, lds_read16(stage, lds_tile_off<D>(row, slot)));   // rows >= rows_here fall outside the SRD
        }
    };

    if (!redo) {
        // O was last written by asm MFMAs: an 8-pass XDL write needs 11+ wait states before v_accvgpr_read may see it
        asm volatile("s_nop 15" ::: "memory");
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 acc[DB];
#pragma unroll
            for (int db = 0; db < DB; ++db) { asm volatile("" : "+a"(oacc[db][qb])); acc[db] = oacc[db][qb]; }
            store_rows(acc, m_run[qb], l_run[qb], wave * 64 + qb * 32, smem + (wave * 2 + qb) * 32 * ROWB);
        }
        return;
    }

    // ---- slow path (rare, data-dependent): exact online softmax with O rescaling, one 32-row q-block per wave at a time ----
    // Plain structure on purpose: one K / V tile in flight, two barriers per tile, builtin MFMAs, everything in VGPRs.
    for (int qb = 0; qb < 2; ++qb) {
        const int row0 = wave * 64 + qb * 32;
        u32x4 qs[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qs[ks] = buf_load16(q_rs, (uint32_t)(row0 + l31) * q_rowb + (2 * ks + hi) * 16);
        f32x16 o[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
        float m = kNegBig, l = 0.f;
        for (int t = 0; t < n_tiles; ++t) {
            __syncthreads();                           // the previous tile's readers are done with slot 0
#pragma unroll
            for (int i = 0; i < DPW; ++i) { dma_k_piece(t, 0, i); dma_v_piece(t, 0, i); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            f32x16 sc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) sc[kb] = LP<T>::mfma(lds_read16(kring, k_rd[ks] + kb * 32 * ROWB), qs[ks], sc[kb]);
            }
            const int n0 = t * kW4BlockN;
            const int lim = CAUSAL ? min(sk - 1, m0 + row0 + l31 + delta) : sk - 1;
            float mxs = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = n0 + 32 * kb + c_row(r, hi);
                    sc[kb][r] = key <= lim ? sc[kb][r] : -INFINITY;
                    mxs = fmaxf(mxs, sc[kb][r]);
                }
            mxs = max_both_halves(mxs);
            const float m_new = fmaxf(m, mxs);
            const float al = fast_exp2((m - m_new) * c);      // m = m_new = kNegBig for rows that still see nothing: exp2(0) = 1
            m = m_new;
            l *= al;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= al;
            const float mc = m * c;
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(__builtin_fmaf(sc[kb][r], c, -mc));
                    ps += pv;
                    sc[kb][r] = pv;
                }
            l += ps;
#pragma unroll
            for (int ts = 0; ts < 4; ++ts) {
                const u32x4 pfr = pack_c_half<T>(sc[ts >> 1], ts & 1);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const u32x2 a0 = lds_read_tr8(vring, v_rd[0][db] + ts * 16 * ROWB);
                    const u32x2 a1 = lds_read_tr8(vring, v_rd[1][db] + ts * 16 * ROWB);
                    o[db] = LP<T>::mfma(u32x4{a0.x, a0.y, a1.x, a1.y}, pfr, o[db]);
                }
            }
        }
        // stage in K ring slots 1 / 2 (the loop above only uses slot 0 of each ring): 8 KiB per wave
        store_rows(o, m, l, row0, smem + TILEB + wave * 32 * ROWB);
    }
}

template <typename T>
static hipError_t launch_w4_t(const FwdKernelParams& kp, hipStream_t stream) {
    const uint32_t grid = kp.varlen_slots != 0 ? kp.varlen_slots * (uint32_t)kp.h : kp.n_q_tiles * (uint32_t)kp.b * (uint32_t)kp.h;
    if (grid == 0) return hipSuccess;
    if (kp.is_causal) hipLaunchKernelGGL((fa_fwd_w4_kernel<T, true>), dim3(grid), dim3(kW4Threads), 0, stream, kp);
    else hipLaunchKernelGGL((fa_fwd_w4_kernel<T, false>), dim3(grid), dim3(kW4Threads), 0, stream, kp);
    return hipGetLastError();
}

hipError_t launch_fwd_w4(FwdKernelParams kp, int dtype, hipStream_t stream) {
    kp.n_q_tiles = (uint32_t)((kp.seqlen_q + kW4BlockM - 1) / kW4BlockM);
    kp.varlen_slots = kp.cu_seqlens_q != nullptr ? varlen_slot_count(kp.total_q, kp.b, kW4BlockM, kp.n_q_tiles) : 0u;
    return dtype == 0 ? launch_w4_t<_Float16>(kp, stream) : launch_w4_t<__bf16>(kp, stream);
}

}  // namespace fa
