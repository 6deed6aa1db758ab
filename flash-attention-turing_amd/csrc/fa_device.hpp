// fa_device.hpp — CDNA4 (gfx950) device-side building blocks shared by the forward and
// backward attention kernels.  Everything here is written for wave64 + MFMA 32x32x16 and the
// 160 KiB LDS of MI355X; there is no other target.
//
// Fragment conventions used throughout (v_mfma_f32_32x32x16_{f16,bf16}, D = A*B + C):
//   A (32 x 16): lane l holds row  i = l & 31, k = 8*(l >> 5) + j, j = 0..7   (8 x 16-bit = 4 VGPRs)
//   B (16 x 32): lane l holds col  n = l & 31, k = 8*(l >> 5) + j
//   C/D (32x32): lane l holds col  n = l & 31, row(r) = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15
// The contraction index k may be permuted freely as long as A and B use the same
// permutation; the kernels exploit that to feed a C-layout accumulator straight back in as a
// B operand (see "k-slot" helpers below) with no LDS round trip and no cross-lane moves.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

#ifndef FA_MFMA16_BUILTIN
#define FA_MFMA16_BUILTIN 0
#endif

namespace fa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short i16x4v __attribute__((__vector_size__(4 * sizeof(short))));

#define FA_DEV __device__ __forceinline__
#define FA_LDS __attribute__((address_space(3)))

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
// Running-max initialiser: finite, so (m_old - m_new) never produces inf - inf = NaN for rows
// that have not seen a visible key yet (SURVEY.md Appendix A dead-row convention).
constexpr float kNegBig = -1.0e30f;

// ---- low-precision element traits ----------------------------------------------------------
template <typename T>
struct LP;

template <>
struct LP<_Float16> {
    static FA_DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    // Accumulate into an AGPR-resident tile (inline asm: "+a").  For kernels whose long-lived
    // accumulators exceed what fits next to the working set in the 256 architectural VGPRs: built
    // with -amdgpu-mfma-vgpr-form every BUILTIN mfma keeps its result in VGPRs (where the softmax
    // VALU can use it directly) while these accumulators, which only MFMAs touch until the
    // epilogue, live in the accumulator half of the register file -- no v_accvgpr shuttling.
    // s_nop 1: a/b may have just been written by VALU (v_cvt_pk) -> MFMA source hazard.
    static FA_DEV void mfma_agpr(f32x16& acc, u32x4 a, u32x4 b) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
    // D(16x16) = A(16x32) * B(32x16) + C: lane l holds A row / B, C column l & 15, k = 8 * (l >> 4) + j, C rows 4 * (l >> 4) + r (tools/probe_isa)
    static FA_DEV f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    // The same, IN PLACE, from inline asm: hipcc does not tie the destination of a 4-pass MFMA to its C operand and allocated new
    // registers for every result (one v_mov_b64 pair per MFMA, +60 registers of pressure, spills).  The caller owns the hazards the
    // compiler cannot see: >= 5 independent MFMAs between two on the same accumulator, wait states before a VALU reads a result.
    static FA_DEV void mfma16_acc(f32x4& acc, u32x4 a, u32x4 b) {
#if FA_MFMA16_BUILTIN
        acc = mfma16(a, b, acc);
#else
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#endif
    }
    static FA_DEV void mfma16_zero(f32x4& acc, u32x4 a, u32x4 b) {      // acc = A * B (C = 0)
#if FA_MFMA16_BUILTIN
        acc = mfma16(a, b, f32x4{0.f, 0.f, 0.f, 0.f});
#else
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
#endif
    }
    // acc = A * B + C with C in its own registers (kept across calls): the forward's scores start from -running_max
    static FA_DEV void mfma16_init(f32x4& acc, u32x4 a, u32x4 b, const f32x4& c) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
    }
    // round-to-nearest-even pack (v_cvt_pk_f16_f32): the reference rounds P/dS/O with
    // cutlass NumericArrayConverter (utils.h:19-27), which is RN as well.
    static FA_DEV uint32_t pack2(float lo, float hi) {
        f32x2 x = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, f16x2));
    }
    static FA_DEV float to_float(uint16_t bits) { return (float)__builtin_bit_cast(_Float16, bits); }
    static constexpr uint32_t kOnes2 = 0x3C003C00u;      // two packed 1.0
    static constexpr uint32_t kBits64 = 0x5400u;         // 64.0
};

template <>
struct LP<__bf16> {
    static FA_DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static FA_DEV void mfma_agpr(f32x16& acc, u32x4 a, u32x4 b) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
    static FA_DEV f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static FA_DEV void mfma16_acc(f32x4& acc, u32x4 a, u32x4 b) {
#if FA_MFMA16_BUILTIN
        acc = mfma16(a, b, acc);
#else
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#endif
    }
    static FA_DEV void mfma16_zero(f32x4& acc, u32x4 a, u32x4 b) {
#if FA_MFMA16_BUILTIN
        acc = mfma16(a, b, f32x4{0.f, 0.f, 0.f, 0.f});
#else
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
#endif
    }
    // acc = A * B + C with C in its own registers (kept across calls): the forward's scores start from -running_max
    static FA_DEV void mfma16_init(f32x4& acc, u32x4 a, u32x4 b, const f32x4& c) {
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
    }
    static FA_DEV uint32_t pack2(float lo, float hi) {
        f32x2 x = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));  // v_cvt_pk_bf16_f32 (RN)
    }
    static FA_DEV float to_float(uint16_t bits) { return __builtin_bit_cast(float, (uint32_t)bits << 16); }
    static constexpr uint32_t kOnes2 = 0x3F803F80u;
    static constexpr uint32_t kBits64 = 0x4280u;
};

// ---- buffer (SRD) addressing -----------------------------------------------------------------
// All global traffic of the tiled kernels goes through raw buffer loads/stores: 32-bit per-lane
// byte offsets against a wave-uniform 64-bit base, and hardware range checking gives the
// reference's predicated copies (utils.h:49-88: zero-fill on read, drop on write) for free.
typedef __amdgpu_buffer_rsrc_t rsrc_t;

FA_DEV rsrc_t make_rsrc(const void* base, uint32_t num_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ (short)0, (int)num_bytes, 0x00020000);
}
FA_DEV u32x4 buf_load16(rsrc_t r, uint32_t byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0); }
FA_DEV void buf_store16(rsrc_t r, uint32_t byte_off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 0); }

// Make a 64-bit pointer provably wave-uniform for the compiler so the descriptor lives in
// SGPRs and no waterfall loop is generated around each buffer op.
template <typename P>
FA_DEV P* uniform_ptr(P* p) {
    uint64_t v = (uint64_t)p;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (P*)(((uint64_t)hi << 32) | lo);
}

// ---- LDS tile layout -------------------------------------------------------------------------
// A staged tile is [rows][D] 16-bit elements, row-major, D*2 bytes per row, with the 16-byte
// slot index XOR-swizzled per row so that BOTH access patterns used by the kernels are bank
// conflict free (banks: (addr/4) % 64 for ds_read_b128 / ds_read_b64_tr_b16):
//   (1) ds_read_b128 "row reads" (MFMA A/B fragments whose 8 k-values are contiguous in the
//       row): a 16-lane service group touches 16 rows distinct mod 16 at one slot index;
//   (2) ds_read_b64_tr_b16 "transposed reads" (fragments whose 8 k-values run DOWN a column):
//       a 32-lane half touches 4 consecutive rows x 64 contiguous bytes.
// swz(row) = ((row & 3) << 2) | ((row >> 2) & 3) is a bijection on row & 15 (fixes 1) whose
// top two bits differ across any 4 consecutive rows aligned to 4 (fixes 2: each of the 4 rows
// lands in a different 64-byte bank quarter).  For D = 64 (128-byte rows, two rows per 256-byte
// bank row) the row parity supplies the missing bit.
template <int D>
FA_DEV uint32_t lds_tile_off(uint32_t row, uint32_t slot /* 16-byte slot in row, 0..D/8-1 */) {
    if constexpr (D == 128) {
        uint32_t s = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
        return row * 256u + (s << 4);
    } else {
        static_assert(D == 64, "head_dim must be 64 or 128");
        // 8 slots per row. bits: slot = (c1 c0 | w) with chunk pairs; use row bits 1..3.
        uint32_t s = slot ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
        return row * 128u + (s << 4);
    }
}

// Inverse of the slot swizzle: which logical 16-byte slot of `row` lives at physical slot `phys`
// (the XOR is an involution).  Used by the LDS-DMA staging, whose LDS image is lane-linear, so the
// swizzle has to be applied on the global SOURCE address instead (cdna guide rule 21).
template <int D>
FA_DEV uint32_t lds_tile_logical_slot(uint32_t row, uint32_t phys) {
    if constexpr (D == 128) return phys ^ (((row & 3) << 2) | ((row >> 2) & 3));
    else return phys ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
}

// Asynchronous HBM -> LDS copy (buffer_load_dwordx4 ... lds): every lane moves 16 bytes from
// rsrc[voffset] to lds_base + 16 * lane (wave-uniform base, lane-linear image); out-of-range
// source addresses write zeros.  Completion is tracked by vmcnt.  gfx950 has the back-off barrier, so
// nothing obliges hipcc to drain vmcnt before s_barrier: callers wait explicitly (s_waitcnt vmcnt(0))
// before the barrier that publishes the data to the other waves.
FA_DEV void dma16_to_lds(rsrc_t r, uint32_t voffset, FA_LDS char* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (FA_LDS void*)lds_base, 16, voffset, 0, 0, 0);
}

// The same copy issued from inline asm so that hipcc does NOT know about it: the compiler treats
// an LDS-DMA it can see as a store that may alias every later LDS read and parks
// "s_waitcnt vmcnt(0)" in front of the next ds_read, which serialises the HBM/L2 latency with
// the matrix phase.  The caller owns the bookkeeping (cdna guide 5.7): one explicit
// "s_waitcnt vmcnt(0)" and a workgroup barrier between this and the first ds_read of the data.
// srd: buffer descriptor words in SGPRs; lds_byte_addr: wave-uniform LDS byte address (M0).
typedef int srd_t __attribute__((ext_vector_type(4)));
FA_DEV srd_t make_srd(const void* base, uint32_t num_bytes) {
    const uint64_t b = (uint64_t)base;
    srd_t s;
    s.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    s.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));   // stride 0, no swizzle
    s.z = __builtin_amdgcn_readfirstlane((int)num_bytes);
    s.w = 0x00020000;                                                              // raw 32-bit data format
    return s;
}
FA_DEV uint32_t lds_addr(const FA_LDS char* p) { return (uint32_t)(uintptr_t)p; }
// SAVE_M0 = false: for kernels that contain NO compiler-visible LDS-DMA (nothing else hipcc generates for them reads M0), where
// the two extra SALU per piece are measurable (0.6 % of the forward); tests/test_kernel_resources_cpu.py checks that such a
// kernel touches M0 only inside these statements.
template <bool SAVE_M0 = true>
FA_DEV void dma16_to_lds_hidden(const srd_t& srd, uint32_t voffset, uint32_t lds_byte_addr) {
    // s_nop 4: SGPRs of the descriptor / M0 source may have just been written by v_readfirstlane
    // or SALU; s_nop 0 after the M0 write (hazard tables 11 / 38).
    // M0 is compiler-reserved (the compiler-visible LDS-DMA builtin addresses LDS through it too) and an
    // "m0" clobber is not honoured for reserved registers, so the statement saves and restores it: a kernel
    // may mix this with dma16_to_lds() without relying on where hipcc happens to re-materialise M0.
    if constexpr (SAVE_M0) {
        uint32_t keep;
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)), "v"(voffset), "s"(srd) : "memory");
    } else {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :: "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)), "v"(voffset), "s"(srd) : "memory");
    }
}

// FOUR consecutive 1-KiB pieces (LDS addresses lds_byte_addr + 0 / 1 / 2 / 3 KiB) in one statement: one descriptor pad for all four, M0 stepped by s_add, and the per-piece
// source offset (wave-uniform tile offset + the lane's swizzled offset) added between the M0 write and its use, where a wait state is needed anyway.
// 12 instructions and 5 wait states instead of 20 + 24 (round 5).  M0 is left at the last piece's address (the SAVE_M0 = false contract above).
FA_DEV void dma16x4_to_lds_hidden(const srd_t& srd, uint32_t tile_off, uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3, uint32_t lds_byte_addr) {
    uint32_t t;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 m0, %1\n\tv_add_u32 %0, %2, %3\n\tbuffer_load_dwordx4 %0, %7, 0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %0, %2, %4\n\tbuffer_load_dwordx4 %0, %7, 0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %0, %2, %5\n\tbuffer_load_dwordx4 %0, %7, 0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %0, %2, %6\n\tbuffer_load_dwordx4 %0, %7, 0 offen lds"
                 : "=&v"(t)
                 : "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)), "s"(__builtin_amdgcn_readfirstlane(tile_off)), "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(srd)
                 : "memory", "scc");
}

// ... and TWO consecutive pieces (fa_fwd_pp.hip: a wave moves two pieces of K and two of V per tile)
FA_DEV void dma16x2_to_lds_hidden(const srd_t& srd, uint32_t tile_off, uint32_t g0, uint32_t g1, uint32_t lds_byte_addr) {
    uint32_t t;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 m0, %1\n\tv_add_u32 %0, %2, %3\n\tbuffer_load_dwordx4 %0, %5, 0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %0, %2, %4\n\tbuffer_load_dwordx4 %0, %5, 0 offen lds"
                 : "=&v"(t)
                 : "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)), "s"(__builtin_amdgcn_readfirstlane(tile_off)), "v"(g0), "v"(g1), "s"(srd)
                 : "memory", "scc");
}

FA_DEV u32x4 lds_read16(const FA_LDS char* base, uint32_t off) { return *(const FA_LDS u32x4*)(base + off); }
FA_DEV void lds_write16(FA_LDS char* base, uint32_t off, u32x4 v) { *(FA_LDS u32x4*)(base + off) = v; }
FA_DEV void lds_write8(FA_LDS char* base, uint32_t off, u32x2 v) { *(FA_LDS u32x2*)(base + off) = v; }

// Hardware transposed read (ds_read_b64_tr_b16). Within each 16-lane group, lane L supplies the
// address of 4 consecutive 16-bit elements; lane L receives, for j = 0..3, element (L & 3) of
// the 8 bytes addressed by lane 4*j + (L >> 2).  With lane L pointing at row (L >> 2), columns
// 4*(L & 3).. of a 4 x 16 block, lane L therefore gets column L of that block, rows 0..3.
FA_DEV u32x2 lds_read_tr8(const FA_LDS char* base, uint32_t off) {
    i16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS i16x4v*)(base + off));
    return __builtin_bit_cast(u32x2, v);
}

// ---- cross-lane helpers ----------------------------------------------------------------------
// value held by lane ^ 32 (the other half-wave owns the other 16 rows of a 32x32 C tile column)
FA_DEV float other_half(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0]: lanes 0-31 keep own, lanes 32-63 receive lanes 0-31; r[1]: lanes 0-31 receive 32-63, 32-63 keep own
    uint32_t o = (threadIdx.x & 32) ? r[0] : r[1];
    return __builtin_bit_cast(float, o);
}
FA_DEV float max_both_halves(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
FA_DEV float sum_both_halves(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}

// the same over the four 16-lane groups of a 16x16x32 fragment (lanes l, l ^ 16, l ^ 32, l ^ 48)
FA_DEV float sum_four_groups(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return sum_both_halves(__builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]));
}
FA_DEV float max_four_groups(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return max_both_halves(fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1])));
}

// element-wise IEEE maximum of three packed 16-bit float pairs (NaN propagates); used on bit patterns, see fa_fwd_pp16.hip
FA_DEV uint32_t pk_max3_f16_bits(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

FA_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
FA_DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
FA_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Compile-time loop: f(integral_constant<int, I>) for I in [B, E).  Arrays indexed this way are scalarised before any loop pass
// runs (a `#pragma unroll` loop over a local array can leave it in scratch when the unroller gets to it late).
template <int B, int E, typename F>
FA_DEV void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Row index (0..31) inside a 32x32 C tile owned by accumulator register r of this lane.
FA_DEV int c_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Pack accumulator registers [8*half .. 8*half+7] of a C tile to 8 low-precision values =
// one MFMA B (or A) fragment whose k-slots are, for this lane (hi = lane >> 5):
//     k-slot j  <->  tile row 16*half + 4*hi + {0,1,2,3,8,9,10,11}[j]
// The matching A operand must be read with the same row set (see tr-read address helpers).
template <typename T>
FA_DEV u32x4 pack_c_half(const f32x16& c, int half) {
    u32x4 o;
    if (half == 0) {
        o.x = LP<T>::pack2(c[0], c[1]);
        o.y = LP<T>::pack2(c[2], c[3]);
        o.z = LP<T>::pack2(c[4], c[5]);
        o.w = LP<T>::pack2(c[6], c[7]);
    } else {
        o.x = LP<T>::pack2(c[8], c[9]);
        o.y = LP<T>::pack2(c[10], c[11]);
        o.z = LP<T>::pack2(c[12], c[13]);
        o.w = LP<T>::pack2(c[14], c[15]);
    }
    return o;
}

// XCD-aware work-item decode.  Blocks are dispatched round-robin over the 8 XCDs
// (block id % 8); every XCD has a private 4 MiB L2.  All tiles that share one (batch, head)
// K/V stream are placed on the same XCD so K/V is fetched from HBM once per XCD, not once per
// workgroup.  Falls back to the plain order when batch*heads is not a multiple of 8.
// group_heads (round 4): the same XCD rule, but the (batch, head) streams of an XCD are dispatched in GROUPS of `group_heads`, tile index first
// within a group.  Under a causal mask the work of a tile grows with its index.  Heaviest-first WITHIN one head (rounds 1-3) is the
// longest-processing-time order only while a head brings at least two workgroups per compute unit (64 query tiles = 16k rows); with fewer, the unit
// that finishes a LIGHT tile first is handed the next head's HEAVIEST one and the chip runs unbalanced (at 1k rows 16 + 16 tile-iterations on some
// units where 20 would do).  Heaviest-first ACROSS all heads balances but lets a head's tiles run at different times, so its K / V is streamed
// from HBM once per tile generation (measured: wins 12-30 % up to 4k rows, loses 6-12 % at 16k).  Groups of ~2 workgroups per compute unit -
// group_heads = ceil(2 * workgroup slots of an XCD / tiles per sequence) - give every unit a heavy and a light tile of the same few heads: balance
// AND locality (tools/sched_sim.py, profiles/r4_causal_tile_order_ab.log, r4_causal_group_order_ab.log).  0 = one head after the other.
FA_DEV void decode_block(uint32_t id, uint32_t tiles_per_bh, uint32_t n_bh, uint32_t& tile, uint32_t& bh, uint32_t group_heads = 0) {
    if ((n_bh & 7u) == 0) {
        uint32_t xcd = id & 7u, slot = id >> 3;
        if (group_heads > 1) {
            const uint32_t per_xcd = n_bh >> 3, span = group_heads * tiles_per_bh;
            const uint32_t grp = slot / span, within = slot - grp * span;
            const uint32_t here = min(group_heads, per_xcd - grp * group_heads);      // the last group may hold fewer heads
            tile = within / here;
            bh = (grp * group_heads + (within - tile * here)) * 8u + xcd;
            return;
        }
        bh = (slot / tiles_per_bh) * 8u + xcd;
        tile = slot % tiles_per_bh;
    } else {
        bh = id / tiles_per_bh;
        tile = id % tiles_per_bh;
    }
}

// ---- varlen: compact grid ------------------------------------------------------------------------------
// The plain varlen grid is tiles(max_seqlen) x batch x heads: a batch of one long and many short sequences launches tens of
// thousands of workgroups that load cu_seqlens and exit (measured: +25 % forward, +20 % backward for 1 x 8192 + 63 x 64 tokens,
// tools/varlen_skew.py).  When the caller passes the packed token count (fa_*_params.total_q / total_k; the torch module passes
// tensor.size(0)), the grid is  slots x heads  with  slots = ceil(total / BM) + batch  >=  sum_i ceil(len_i / BM),  and every
// workgroup finds its (sequence, tile) here.  Cost: ONE round of independent loads + a wave prefix sum, whatever the batch size
// (up to kVarlenMaxBatch sequences = 64 lanes x 8; larger batches keep the plain grid).  Every wave of the workgroup computes the
// same answer from the same data, so all waves agree on exiting.
constexpr int kVarlenSeqPerLane = 8;
constexpr int kVarlenMaxBatch = 64 * kVarlenSeqPerLane;

inline uint32_t varlen_slot_count(int64_t total, int b, int bm, uint32_t tiles_max_seq) {   // host side (launchers)
    if (total <= 0 || b > kVarlenMaxBatch) return 0;
    const uint64_t compact = (uint64_t)((total + bm - 1) / bm) + (uint64_t)b, plain = (uint64_t)tiles_max_seq * (uint64_t)b;
    return compact < plain ? (uint32_t)compact : 0;      // never a larger grid than the plain one
}

// slot -> (seq, tile inside seq, tiles of seq); false = a slack slot past the last tile (exit)
template <int BM>
FA_DEV bool varlen_slot_lookup(const int32_t* cu, int b, uint32_t slot, int& seq, int& tile, int& ntiles) {
    const int lane = threadIdx.x & 63;
    const int per = (b + 63) >> 6;                               // sequences per lane, <= kVarlenSeqPerLane
    const int i0 = lane * per;
    int c[kVarlenSeqPerLane + 1];
#pragma unroll
    for (int j = 0; j <= kVarlenSeqPerLane; ++j) c[j] = cu[min(i0 + min(j, per), b)];   // independent loads: one latency
    uint32_t t[kVarlenSeqPerLane], mine = 0;
#pragma unroll
    for (int j = 0; j < kVarlenSeqPerLane; ++j) {
        t[j] = (j < per) ? (uint32_t)((c[j + 1] - c[j] + BM - 1) / BM) : 0u;
        mine += t[j];
    }
    uint32_t incl = mine;                                        // inclusive prefix over the 64 lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += y;
    }
    uint32_t run = incl - mine;
    int f_seq = -1, f_tile = 0, f_nt = 0;
#pragma unroll
    for (int j = 0; j < kVarlenSeqPerLane; ++j) {
        if (slot >= run && slot < run + t[j]) { f_seq = i0 + j; f_tile = (int)(slot - run); f_nt = (int)t[j]; }
        run += t[j];
    }
    const uint64_t m = __ballot(f_seq >= 0);
    if (m == 0) return false;
    const int src = __ffsll((long long)m) - 1;
    seq = __builtin_amdgcn_readlane(f_seq, src);
    tile = __builtin_amdgcn_readlane(f_tile, src);
    ntiles = __builtin_amdgcn_readlane(f_nt, src);
    return true;
}

// The same lookup for CAUSAL launches, heaviest items first (round 4): under the mask the work of a tile grows with its index (query tiles; key
// blocks run the other way: block 0 is the heaviest), and a packed batch is dispatched well only if the heavy tiles of ALL sequences go first
// (longest-processing-time order; one sequence after the other hands the compute unit that finishes a light tile the next sequence's heaviest).
// Items are ordered by their distance L from the light end of their sequence, descending, ties by sequence index.  `tile` comes back as
// ntiles - 1 - L: the forward / dQ kernels reverse it to L (heaviest query tile first), dK/dV takes it as is (key block 0 = L = ntiles - 1 first).
// One sequence per lane (b <= 64), at most `tiles_max` <= 64 tiles per sequence (sequences longer than the declared max_seqlen lose their
// surplus tiles here, which exit immediately in the kernels anyway).  Cost: a scalar loop over the levels above the item's own
// (ballot + popcount per level), ~30 cycles each.
constexpr uint32_t kVarlenHeavyFirst = 0xffffffffu;      // value of the kernel parameter `group_heads` that selects this lookup on a compact grid
template <int BM>
FA_DEV bool varlen_slot_lookup_heavy_first(const int32_t* cu, int b, uint32_t tiles_max, uint32_t slot, int& seq, int& tile, int& ntiles) {
    const int lane = threadIdx.x & 63;
    const int c0 = cu[min(lane, b)], c1 = cu[min(lane + 1, b)];
    const int t = lane < b ? min((c1 - c0 + BM - 1) / BM, (int)tiles_max) : 0;
    uint32_t before = 0;
    for (int L = (int)tiles_max - 1; L >= 0; --L) {
        const uint64_t own = __ballot(t > L);                                   // sequences that have an item at level L
        const uint32_t c = (uint32_t)__popcll((unsigned long long)own);
        if (slot < before + c) {
            const int r = (int)(slot - before);                                   // the r-th of them, in index order
            const int below = __popcll((unsigned long long)(own & ((1ull << lane) - 1ull)));
            const uint64_t pick = __ballot(((own >> lane) & 1ull) != 0 && below == r);
            seq = __ffsll((long long)pick) - 1;
            ntiles = __builtin_amdgcn_readlane(t, seq);
            tile = ntiles - 1 - L;
            return true;
        }
        before += c;
    }
    return false;                                                                 // a slack slot past the last item
}

// Common block decode: plain grid (tiles_per_bh x batch x heads, XCD-aware) or compact varlen grid (slots x heads).
// `tile` comes back un-reversed; `ntiles` = tiles of THIS sequence (compact) or tiles_per_bh (plain) for the causal reversal.
template <int BM>
FA_DEV bool decode_work(uint32_t id, uint32_t tiles_per_bh, uint32_t slots, const int32_t* cu, int b, int nheads,
                        int& tile, int& batch, int& head, int& ntiles, uint32_t group_heads = 0) {
    if (slots != 0) {
        uint32_t slot, hd;
        if (group_heads == kVarlenHeavyFirst) {                    // causal packed batch: heaviest items first, across sequences AND across heads
            decode_block(id, slots, (uint32_t)nheads, slot, hd, (uint32_t)nheads >> 3);
            head = (int)hd;
            return varlen_slot_lookup_heavy_first<BM>(cu, b, tiles_per_bh, slot, batch, tile, ntiles);
        }
        decode_block(id, slots, (uint32_t)nheads, slot, hd);
        head = (int)hd;
        return varlen_slot_lookup<BM>(cu, b, slot, batch, tile, ntiles);
    }
    uint32_t t, bh;
    decode_block(id, tiles_per_bh, (uint32_t)(b * nheads), t, bh, group_heads);
    tile = (int)t; batch = (int)(bh / (uint32_t)nheads); head = (int)(bh % (uint32_t)nheads); ntiles = (int)tiles_per_bh;
    return true;
}

}  // namespace fa
