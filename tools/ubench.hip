// ubench.hip — instruction-cost microbenchmarks on one CU of an MI355X (gfx950), used to size
// the softmax (VALU) phase against the matrix phase of the attention kernels.
// Each test: one workgroup; every wave runs ITERS x 16 independent copies of one instruction;
// wave 0 reports s_memtime cycles / instruction.  "pair" tests put two waves on each SIMD
// (512 threads): waves 0-3 run body A, waves 4-7 body B, both timed.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ITERS 2000
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { T_EXP, T_FMA, T_ADD, T_MAX3, T_CVT, T_PKFMA, T_PKMUL, T_PKADD, T_EXPH, T_LDEXP, T_LSHLADD, T_MIX_EXP_FMA, T_MFMA, T_LOG, T_RCP,
       T_MFMA16, T_STREAM4, T_STREAM5, T_STREAM6, T_MFMA_NOP20, T_MFMA_NOP26, T_MFMA_SLEEP, T_MN3, T_MN4, T_MN5, T_MN6, T_MN7, T_NUM };
static const char* kNames[] = {"v_exp_f32", "v_fma_f32", "v_add_f32", "v_max3_f32", "v_cvt_pk_f16_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                               "v_pk_add_f32", "v_exp_f16", "v_ldexp_f32", "v_lshl_add_u32", "8 exp + 8 fma interleaved", "v_mfma_32x32x16_f16",
                               "v_log_f32", "v_rcp_f32", "v_mfma_16x16x32_f16", "stream 1 mfma : 4 valu (1 exp)", "stream 1 mfma : 5 valu (1 exp)", "stream 1 mfma : 6 valu (1 exp)", "mfma + s_nop(20)", "mfma + s_nop(26)", "mfma + s_sleep 0", "mn3", "mn4", "mn5", "mn6", "mn7"};

template <int TEST>
__device__ __forceinline__ void body(float (&r)[16], f32x2 (&pk)[16], f32x16& acc, f32x16& acc2, f16x8 a, f16x8 b) {
    if constexpr (TEST == T_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_LOG) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_CVT) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(pk[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(pk[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(pk[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_EXPH) {
#define X(i) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_LDEXP) {
#define X(i) asm volatile("v_ldexp_f32 %0, %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_LSHLADD) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_MIX_EXP_FMA) {
#define X(i) if ((i) & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])); else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_MFMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
        }
    } else if constexpr (TEST == T_STREAM4 || TEST == T_STREAM5 || TEST == T_STREAM6) {
        // 16 x { 1 MFMA (4 rotating accumulators would need more regs: use 2), 1 v_exp, 3..5 plain VALU }
#define X(i)                                                                                         \
        if ((i) & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);               \
        else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);                      \
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));                                               \
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[(i + 5) & 15]));                            \
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[(i + 9) & 15]));                                \
        asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(r[(i + 3) & 15]));                           \
        if (TEST != T_STREAM4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[(i + 7) & 15]));  \
        if (TEST == T_STREAM6) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[(i + 11) & 15]));
        REP16(X)
#undef X
    } else if constexpr (TEST == T_MFMA_NOP20 || TEST == T_MFMA_NOP26 || TEST == T_MFMA_SLEEP) {
        // does an MFMA wave that does NOT ask for the VALU/MFMA issue port early leave it to its SIMD partner?
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            if (TEST == T_MFMA_NOP20) asm volatile("s_nop 15\n\ts_nop 3");
            else if (TEST == T_MFMA_NOP26) asm volatile("s_nop 15\n\ts_nop 9");
            else asm volatile("s_sleep 0");
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
            if (TEST == T_MFMA_NOP20) asm volatile("s_nop 15\n\ts_nop 3");
            else if (TEST == T_MFMA_NOP26) asm volatile("s_nop 15\n\ts_nop 9");
            else asm volatile("s_sleep 0");
        }
    } else if constexpr (TEST >= T_MN3 && TEST <= T_MN7) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            if (TEST == T_MN3) asm volatile("s_nop 3"); else if (TEST == T_MN4) asm volatile("s_nop 4"); else if (TEST == T_MN5) asm volatile("s_nop 5");
            else if (TEST == T_MN6) asm volatile("s_nop 6"); else asm volatile("s_nop 7");
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
            if (TEST == T_MN3) asm volatile("s_nop 3"); else if (TEST == T_MN4) asm volatile("s_nop 4"); else if (TEST == T_MN5) asm volatile("s_nop 5");
            else if (TEST == T_MN6) asm volatile("s_nop 6"); else asm volatile("s_nop 7");
        }
    } else if constexpr (TEST == T_MFMA16) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 c0 = {acc[0], acc[1], acc[2], acc[3]}, c1 = {acc[4], acc[5], acc[6], acc[7]};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        }
        acc[0] = c0[0]; acc[4] = c1[0];
    }
}

template <int TA, int TB, int PA = 0, int PB = 0>
__global__ void k_bench(unsigned long long* out, float seed) {
    float r[16];
    f32x2 pk[16];
    f32x16 acc, acc2;
    f16x8 a, b;
    for (int i = 0; i < 16; ++i) { r[i] = seed + threadIdx.x * 1e-3f + i; pk[i] = f32x2{r[i], r[i] + 1.f}; acc[i] = r[i]; acc2[i] = r[i]; }
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    const int wave = threadIdx.x >> 6;
    if (wave < 4) __builtin_amdgcn_s_setprio(PA); else __builtin_amdgcn_s_setprio(PB);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < ITERS; ++it) body<TA>(r, pk, acc, acc2, a, b);
    } else {
        for (int it = 0; it < ITERS; ++it) body<TB>(r, pk, acc, acc2, a, b);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sink = 0.f;
    for (int i = 0; i < 16; ++i) sink += r[i] + pk[i].x + pk[i].y + acc[i] + acc2[i];
    if (sink == 12345.678f) out[100] = 1;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}


// ---- round 6: the cadence of v_mfma_f32_16x16x32 issued by ONE wave -------------------------------------------------------------------
//   A wave's stream: NACC independent accumulators (inline asm, in place, like the kernels), and per PAIR of MFMAs one of
//   C_NONE nothing, C_B128 one ds_read_b128 (never waited on), C_B128W the same + s_waitcnt lgkmcnt(1), C_TRW two ds_read_b64_tr_b16 + s_waitcnt lgkmcnt(2)
//   (the forward's P.V fragment step as hipcc emits it), C_NOP one s_nop 0, C_TR2W = C_TRW with ONE wait per two pairs, C_SOFTMAX = no MFMAs: 3 VALU of the softmax mix per "pair".
// The figure printed is cycles per MFMA (16 = the pipe's rate); for C_SOFTMAX cycles per VALU instruction.
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
enum { C_NONE, C_B128, C_B128W, C_TRW, C_NOP, C_TR2W, C_SOFTMAX, C_IDLE, C_FMA, C_PKFMA, C_EXP, C_CVT, C_SOFTMAX_PK };
template <int NACC, int KIND, bool AGPR>
__device__ __forceinline__ void cad_stream(f32x4_ (&c)[16], float (&r)[16], u32x4_& fr, unsigned int la, f16x8 a, f16x8 b, int iters) {
    if constexpr (KIND == C_IDLE) return;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == C_FMA || KIND == C_PKFMA || KIND == C_EXP || KIND == C_CVT || KIND == C_SOFTMAX_PK) {
            // single-instruction streams, and the softmax mix with PACKED multiply-subtracts: per 16 instructions 4 v_pk_fma_f32 (= 8 scores), 8 v_exp_f32, 4 v_cvt_pk
            // (the work of 20 instructions of the scalar mix)
#define X(i)                                                                                                                        \
            if (KIND == C_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));                                                  \
            else if (KIND == C_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));                                                   \
            else if (KIND == C_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[i]));                                        \
            else if (KIND == C_PKFMA || (i) % 4 == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(f32x2*)&r[((i) & 7) * 2]));     \
            else if ((i) % 4 == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[i]));                                         \
            else asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == C_SOFTMAX) {
#define X(i)                                                                                                    \
            if ((i) % 3 == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));                             \
            else if ((i) % 3 == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));                               \
            else if ((i) == 5) asm volatile("v_pk_maximum3_f16 %0, %0, %0, %0" : "+v"(r[i]));                    \
            else asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[i]));
            REP16(X)
#undef X
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[i % NACC]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i % NACC]) : "v"(a), "v"(b));
                if (i & 1) {
                    if constexpr (KIND == C_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(fr) : "v"(la));
                    if constexpr (KIND == C_B128W) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(1)" : "=v"(fr) : "v"(la));
                    if constexpr (KIND == C_TRW || KIND == C_TR2W) {
                        unsigned long long t0_, t1_;
                        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096" : "=v"(t0_), "=v"(t1_) : "v"(la));
                        if (KIND == C_TRW || (i & 2)) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                        fr[0] = (unsigned int)t0_; fr[2] = (unsigned int)t1_;
                    }
                    if constexpr (KIND == C_NOP) asm volatile("s_nop 0");
                }
            }
        }
    }
}
template <int NA, int KA, bool AGA, int NB, int KB, int PA, int PB>
__global__ void k_cad(unsigned long long* out, float seed) {
    __shared__ unsigned int lds_[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds_[i] = 0x3c003c00u;
    f32x4_ c[16];
    float r[16];
    f16x8 a, b;
    for (int i = 0; i < 16; ++i) { r[i] = seed + threadIdx.x * 1e-3f + i; c[i] = f32x4_{r[i], r[i], r[i], r[i]}; }
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    u32x4_ fr = {0, 0, 0, 0};
    const unsigned int la = (unsigned int)(size_t)lds_ + (threadIdx.x & 63) * 16;
    const int wave = threadIdx.x >> 6;
    if (wave < 4) __builtin_amdgcn_s_setprio(PA); else __builtin_amdgcn_s_setprio(PB);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) cad_stream<NA, KA, AGA>(c, r, fr, la, a, b, ITERS);
    else cad_stream<NB, KB, false>(c, r, fr, la, a, b, ITERS);
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float sink = __builtin_bit_cast(float, fr[0]) + __builtin_bit_cast(float, fr[2]);
    for (int i = 0; i < 16; ++i) { if constexpr (AGA) asm volatile("" : "+a"(c[i])); else asm volatile("" : "+v"(c[i])); sink += r[i] + c[i][0] + c[i][3]; }
    if (sink == 12345.678f) out[100] = 1;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int TA, int TB, int PA = 0, int PB = 0>
void run(unsigned long long* d, int threads, const char* label) {
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        k_bench<TA, TB, PA, PB><<<1, threads>>>(d, 1.0f);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const double n = (double)ITERS * 16;
    if (threads == 256) printf("%-44s alone: %6.2f cyc/instr/wave\n", label, h[0] / n);
    else printf("%-44s A: %6.2f cyc/instr   B: %6.2f cyc/instr (waves 0 and 4 share SIMD0)\n", label, h[0] / n, h[4] / n);
}

template <int NA, int KA, bool AGA, int NB, int KB, int PA = 0, int PB = 0>
void run_cad(unsigned long long* d, int threads, const char* label) {
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        k_cad<NA, KA, AGA, NB, KB, PA, PB><<<1, threads>>>(d, 1.0f);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const double n = (double)ITERS * 16;
    if (threads == 256) printf("%-64s alone: %6.2f cyc per MFMA (or VALU)\n", label, h[0] / n);
    else printf("%-64s A: %6.2f   B: %6.2f  cyc per MFMA / VALU (waves 0 and 4 share SIMD0)\n", label, h[0] / n, h[4] / n);
}

int main() {
    unsigned long long* d;
    CK(hipMalloc(&d, 1024));
    printf("== one wave per SIMD ==\n");
    run<T_EXP, T_EXP>(d, 256, kNames[T_EXP]);
    run<T_LOG, T_LOG>(d, 256, kNames[T_LOG]);
    run<T_RCP, T_RCP>(d, 256, kNames[T_RCP]);
    run<T_EXPH, T_EXPH>(d, 256, kNames[T_EXPH]);
    run<T_FMA, T_FMA>(d, 256, kNames[T_FMA]);
    run<T_ADD, T_ADD>(d, 256, kNames[T_ADD]);
    run<T_MAX3, T_MAX3>(d, 256, kNames[T_MAX3]);
    run<T_CVT, T_CVT>(d, 256, kNames[T_CVT]);
    run<T_PKFMA, T_PKFMA>(d, 256, kNames[T_PKFMA]);
    run<T_PKMUL, T_PKMUL>(d, 256, kNames[T_PKMUL]);
    run<T_PKADD, T_PKADD>(d, 256, kNames[T_PKADD]);
    run<T_LDEXP, T_LDEXP>(d, 256, kNames[T_LDEXP]);
    run<T_LSHLADD, T_LSHLADD>(d, 256, kNames[T_LSHLADD]);
    run<T_MIX_EXP_FMA, T_MIX_EXP_FMA>(d, 256, kNames[T_MIX_EXP_FMA]);
    run<T_MFMA, T_MFMA>(d, 256, kNames[T_MFMA]);
    run<T_MFMA16, T_MFMA16>(d, 256, kNames[T_MFMA16]);
    printf("== two waves per SIMD (A = waves 0-3, B = waves 4-7) ==\n");
    run<T_EXP, T_EXP>(d, 512, "exp | exp");
    run<T_FMA, T_FMA>(d, 512, "fma | fma");
    run<T_EXP, T_FMA>(d, 512, "exp | fma");
    run<T_MFMA, T_MFMA>(d, 512, "mfma | mfma");
    run<T_MFMA, T_EXP>(d, 512, "mfma | exp");
    run<T_MFMA, T_FMA>(d, 512, "mfma | fma");
    run<T_MFMA, T_ADD>(d, 512, "mfma | add");
    run<T_MFMA, T_CVT>(d, 512, "mfma | cvt_pk");
    run<T_MFMA, T_MAX3>(d, 512, "mfma | max3");
    run<T_MFMA, T_PKFMA>(d, 512, "mfma | pk_fma");
    run<T_MFMA, T_MIX_EXP_FMA>(d, 512, "mfma | exp+fma mix");
    run<T_MFMA16, T_EXP>(d, 512, "mfma16 | exp");
    run<T_STREAM4, T_STREAM4>(d, 256, "mixed stream 1:4 alone (cyc per 1/16 body)");
    run<T_STREAM5, T_STREAM5>(d, 256, "mixed stream 1:5 alone");
    run<T_STREAM6, T_STREAM6>(d, 256, "mixed stream 1:6 alone");
    run<T_STREAM4, T_STREAM4>(d, 512, "mixed 1:4 | mixed 1:4");
    run<T_STREAM5, T_STREAM5>(d, 512, "mixed 1:5 | mixed 1:5");
    run<T_STREAM6, T_STREAM6>(d, 512, "mixed 1:6 | mixed 1:6");
    run<T_MFMA_NOP20, T_MFMA_NOP20>(d, 256, "mfma + s_nop 20 alone");
    run<T_MFMA_NOP26, T_MFMA_NOP26>(d, 256, "mfma + s_nop 26 alone");
    run<T_MFMA_NOP20, T_FMA>(d, 512, "mfma+nop20 | fma");
    run<T_MFMA_NOP26, T_FMA>(d, 512, "mfma+nop26 | fma");
    run<T_MFMA_NOP26, T_EXP>(d, 512, "mfma+nop26 | exp");
    run<T_MFMA_NOP26, T_MIX_EXP_FMA>(d, 512, "mfma+nop26 | exp+fma");
    run<T_MFMA_SLEEP, T_FMA>(d, 512, "mfma+sleep0 | fma");
    run<T_MN3, T_MIX_EXP_FMA>(d, 512, "mfma+s_nop 3 | exp+fma");
    run<T_MN4, T_MIX_EXP_FMA>(d, 512, "mfma+s_nop 4 | exp+fma");
    run<T_MN5, T_MIX_EXP_FMA>(d, 512, "mfma+s_nop 5 | exp+fma");
    run<T_MN6, T_MIX_EXP_FMA>(d, 512, "mfma+s_nop 6 | exp+fma");
    run<T_MN7, T_MIX_EXP_FMA>(d, 512, "mfma+s_nop 7 | exp+fma");
    run<T_MN5, T_MN5>(d, 256, "mfma+s_nop 5 alone");
    run<T_MN6, T_MN6>(d, 256, "mfma+s_nop 6 alone");
    printf("== round 6: cadence of v_mfma_f32_16x16x32 from ONE wave per SIMD (cycles per MFMA; the pipe needs 16) ==\n");
    run_cad<2, C_NONE, false, 8, C_IDLE>(d, 256, "mfma16, 2 accumulators");
    run_cad<4, C_NONE, false, 8, C_IDLE>(d, 256, "mfma16, 4 accumulators");
    run_cad<8, C_NONE, false, 8, C_IDLE>(d, 256, "mfma16, 8 accumulators");
    run_cad<16, C_NONE, false, 8, C_IDLE>(d, 256, "mfma16, 16 accumulators");
    run_cad<8, C_NONE, true, 8, C_IDLE>(d, 256, "mfma16, 8 accumulators in AGPRs");
    run_cad<8, C_B128, false, 8, C_IDLE>(d, 256, "mfma16 8 acc + ds_read_b128 per pair");
    run_cad<8, C_B128W, false, 8, C_IDLE>(d, 256, "mfma16 8 acc + (ds_read_b128 + s_waitcnt) per pair");
    run_cad<8, C_TRW, false, 8, C_IDLE>(d, 256, "mfma16 8 acc + (2 ds_read_tr + s_waitcnt) per pair");
    run_cad<8, C_TR2W, false, 8, C_IDLE>(d, 256, "mfma16 8 acc + 2 ds_read_tr per pair, one s_waitcnt per 2 pairs");
    run_cad<8, C_NOP, false, 8, C_IDLE>(d, 256, "mfma16 8 acc + s_nop 0 per pair");
    run_cad<8, C_SOFTMAX, false, 8, C_IDLE>(d, 256, "softmax VALU mix (fma, exp, cvt / max3)");
    printf("== round 6: the MFMA wave (A) sharing its SIMD with a softmax-mix wave (B) ==\n");
    run_cad<8, C_NONE, false, 8, C_SOFTMAX>(d, 512, "mfma16 | softmax mix");
    run_cad<8, C_NONE, false, 8, C_SOFTMAX, 1, 0>(d, 512, "mfma16 (p1) | softmax mix (p0)");
    run_cad<8, C_NONE, false, 8, C_SOFTMAX, 0, 1>(d, 512, "mfma16 (p0) | softmax mix (p1)");
    run_cad<8, C_NONE, false, 8, C_SOFTMAX, 3, 0>(d, 512, "mfma16 (p3) | softmax mix (p0)");
    run_cad<8, C_NONE, true, 8, C_SOFTMAX>(d, 512, "mfma16 AGPR | softmax mix");
    run_cad<8, C_B128W, false, 8, C_SOFTMAX>(d, 512, "mfma16 + (b128 + wait) per pair | softmax mix");
    run_cad<8, C_TRW, false, 8, C_SOFTMAX>(d, 512, "mfma16 + (2 tr + wait) per pair | softmax mix");
    run_cad<8, C_TRW, false, 8, C_SOFTMAX, 1, 0>(d, 512, "mfma16 + (2 tr + wait) per pair (p1) | softmax mix (p0)");
    run_cad<8, C_TR2W, false, 8, C_SOFTMAX>(d, 512, "mfma16 + 2 tr per pair, wait per 2 pairs | softmax mix");
    run_cad<8, C_SOFTMAX, false, 8, C_TRW>(d, 512, "softmax mix | mfma16 + (2 tr + wait) per pair  [VALU wave older]");
    run_cad<8, C_FMA, false, 8, C_IDLE>(d, 256, "v_fma_f32 stream");
    run_cad<8, C_PKFMA, false, 8, C_IDLE>(d, 256, "v_pk_fma_f32 stream");
    run_cad<8, C_EXP, false, 8, C_IDLE>(d, 256, "v_exp_f32 stream");
    run_cad<8, C_SOFTMAX_PK, false, 8, C_IDLE>(d, 256, "softmax mix with packed fma (4 pk_fma, 8 exp, 4 cvt)");
    run_cad<8, C_NONE, false, 8, C_FMA>(d, 512, "mfma16 | v_fma_f32");
    run_cad<8, C_NONE, false, 8, C_PKFMA>(d, 512, "mfma16 | v_pk_fma_f32");
    run_cad<8, C_NONE, false, 8, C_EXP>(d, 512, "mfma16 | v_exp_f32");
    run_cad<8, C_NONE, false, 8, C_CVT>(d, 512, "mfma16 | v_cvt_pk_f16_f32");
    run_cad<8, C_NONE, false, 8, C_SOFTMAX_PK>(d, 512, "mfma16 | softmax mix with packed fma");
    run_cad<8, C_TRW, false, 8, C_FMA>(d, 512, "mfma16 + (2 tr + wait) per pair | v_fma_f32");
    run_cad<8, C_TRW, false, 8, C_PKFMA>(d, 512, "mfma16 + (2 tr + wait) per pair | v_pk_fma_f32");
    run_cad<8, C_TRW, false, 8, C_EXP>(d, 512, "mfma16 + (2 tr + wait) per pair | v_exp_f32");
    run_cad<8, C_TRW, false, 8, C_SOFTMAX_PK>(d, 512, "mfma16 + (2 tr + wait) per pair | softmax mix with packed fma");
    run_cad<8, C_NONE, false, 8, C_NONE>(d, 512, "mfma16 | mfma16");
    run_cad<8, C_TRW, false, 8, C_TRW>(d, 512, "mfma16 + (2 tr + wait) | the same");
    printf("== priorities (A prio, B prio) ==\n");
    run<T_MFMA, T_FMA, 0, 1>(d, 512, "mfma(p0) | fma(p1)");
    run<T_MFMA, T_FMA, 0, 3>(d, 512, "mfma(p0) | fma(p3)");
    run<T_MFMA, T_FMA, 1, 0>(d, 512, "mfma(p1) | fma(p0)");
    run<T_MFMA, T_EXP, 0, 1>(d, 512, "mfma(p0) | exp(p1)");
    run<T_MFMA, T_EXP, 0, 3>(d, 512, "mfma(p0) | exp(p3)");
    run<T_MFMA, T_MIX_EXP_FMA, 0, 1>(d, 512, "mfma(p0) | exp+fma(p1)");
    run<T_FMA, T_MFMA, 1, 0>(d, 512, "fma(p1) | mfma(p0)  [VALU wave older]");
    run<T_FMA, T_MFMA, 0, 0>(d, 512, "fma(p0) | mfma(p0)  [VALU wave older]");
    run<T_EXP, T_MFMA, 0, 0>(d, 512, "exp(p0) | mfma(p0)  [VALU wave older]");
    run<T_MIX_EXP_FMA, T_MFMA, 0, 0>(d, 512, "exp+fma(p0) | mfma(p0)  [VALU wave older]");
    return 0;
}
