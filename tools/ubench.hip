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
