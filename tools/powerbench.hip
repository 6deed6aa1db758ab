// powerbench.hip -- what the power-limited matrix pipes of an MI355X deliver as a function of the MFMA variant and of the DATA:
// chip-wide dependency-free MFMA loops (2 waves / SIMD, 4 accumulators per wave), every variant 5x interleaved.  The attention
// kernels run against a clock that falls as the pipes fill (profiles/NOTEBOOK.md 3c); this probe asks which knobs move that ceiling at all.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// VARIANT 0: 32x32x16 f16   1: 16x16x32 f16   2: 32x32x16 bf16   3: 16x16x32 bf16
template <int VARIANT>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, const float* src, int iters) {
    f16x8 ah, bh; bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) {
        const float x = src[(threadIdx.x * 8 + i) & 4095], y = src[(threadIdx.x * 8 + i + 77) & 4095];
        ah[i] = (_Float16)x; bh[i] = (_Float16)y; ab[i] = (__bf16)x; bb[i] = (__bf16)y;
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0}, d4 = {0}, d5 = {0}, d6 = {0}, d7 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (VARIANT == 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
            } else if constexpr (VARIANT == 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
            } else if constexpr (VARIANT == 1) {     // 8 x (16x16x32) = the FLOPs of 4 x (32x32x16)
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d3, 0, 0, 0);
                d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d4, 0, 0, 0); d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d5, 0, 0, 0);
                d6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d6, 0, 0, 0); d7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d7, 0, 0, 0);
            } else {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d3, 0, 0, 0);
                d4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d4, 0, 0, 0); d5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d5, 0, 0, 0);
                d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d6, 0, 0, 0); d7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d7, 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i] + d4[i] + d5[i] + d6[i] + d7[i];
    if (s == 1234.5f) out[1] = 1;
}
#define VALU4(a, b, c, d) if constexpr (WITH_VALU) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// the forward kernel's mix per 32768 FLOP: 4 VALU + 1 KiB of LDS reads, around one 32x32x16 or two 16x16x32 MFMAs (fp16, N(0,1) operands)
template <int SMALL, int NLDS = 4, int WITH_VALU = 1>
__global__ __launch_bounds__(512, 2) void kmix(unsigned long long* out, const float* src, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const unsigned int lbase = (unsigned int)(size_t)(lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 256);
    u32x4 ld[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)src[(threadIdx.x * 8 + i) & 4095]; bh[i] = (_Float16)src[(threadIdx.x * 8 + i + 77) & 4095]; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0}, d4 = {0}, d5 = {0}, d6 = {0}, d7 = {0};
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = (float)ah[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (!SMALL) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0); VALU4(r[0], r[1], r[2], r[3]);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0); VALU4(r[4], r[5], r[6], r[7]);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0); VALU4(r[0], r[1], r[2], r[3]);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0); VALU4(r[4], r[5], r[6], r[7]);
            } else {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d1, 0, 0, 0); VALU4(r[0], r[1], r[2], r[3]);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d3, 0, 0, 0); VALU4(r[4], r[5], r[6], r[7]);
                d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d4, 0, 0, 0); d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d5, 0, 0, 0); VALU4(r[0], r[1], r[2], r[3]);
                d6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d6, 0, 0, 0); d7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d7, 0, 0, 0); VALU4(r[4], r[5], r[6], r[7]);
            }
#pragma unroll
            for (int q = 0; q < NLDS; ++q) {
                asm volatile("" :: "v"(ld[q & 3]));
                const unsigned int a0 = lbase + (((it * 4 + j + q * 7) & 7) * 8192u);
                asm volatile("ds_read_b128 %0, %1" : "=v"(ld[q & 3]) : "v"(a0));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i] + d4[i] + d5[i] + d6[i] + d7[i];
    for (int i = 0; i < 8; ++i) s += r[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 lacc = ld[0] ^ ld[1] ^ ld[2] ^ ld[3];
    if (s == 1234.5f || lacc.x + lacc.y + lacc.z + lacc.w == 77u) out[1] = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef void (*kfn)(unsigned long long*, const float*, int);
static double time_one(kfn f, unsigned long long* d, const float* src) {
    const int iters = 20000, threads = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(f, dim3(256), dim3(threads), 0, 0, d, src, 100); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(f, dim3(256), dim3(threads), 0, 0, d, src, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 256.0 * (threads / 64) * iters * 16.0 * 2 * 32 * 32 * 16 / (ms * 1e9);
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    float h[3][4096]; srand(3);
    for (int i = 0; i < 4096; ++i) {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
        h[0][i] = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);   // N(0,1): what bench.py feeds the kernels
        h[1][i] = 0.f;                                             // zeros: no toggling at all
        h[2][i] = (rand() & 1) ? 1.0f : 0.5f;                      // two exact values: mantissas all zero, exponent bit toggles
    }
    float* src[3];
    for (int k_ = 0; k_ < 3; ++k_) { CK(hipMalloc(&src[k_], sizeof(h[0]))); CK(hipMemcpy(src[k_], h[k_], sizeof(h[0]), hipMemcpyHostToDevice)); }
    const char* dn[3] = {"N(0,1) data", "zeros", "{0.5, 1.0}"};
    struct V { const char* label; kfn f; } vs[] = {{"32x32x16 f16", k<0>}, {"16x16x32 f16", k<1>}, {"32x32x16 bf16", k<2>}, {"16x16x32 bf16", k<3>}};
    double tf[4][3][5];
    for (int r = 0; r < 5; ++r) for (int v = 0; v < 4; ++v) for (int k_ = 0; k_ < 3; ++k_) tf[v][k_][r] = time_one(vs[v].f, d, src[k_]);
    printf("%-16s %-14s %8s %8s %8s   (TFLOP/s, 5 interleaved runs; dense fp16 / bf16 peak 2500)\n", "mfma", "operands", "min", "median", "max");
    for (int v = 0; v < 4; ++v) for (int k_ = 0; k_ < 3; ++k_) {
        double* t = tf[v][k_];
        for (int a = 0; a < 5; ++a) for (int b = a + 1; b < 5; ++b) if (t[b] < t[a]) { double x = t[a]; t[a] = t[b]; t[b] = x; }
        printf("%-16s %-14s %8.0f %8.0f %8.0f\n", vs[v].label, dn[k_], t[0], t[2], t[4]);
    }
    double tm[2][5];
    for (int r = 0; r < 5; ++r) { tm[0][r] = time_one(kmix<0>, d, src[0]); tm[1][r] = time_one(kmix<1>, d, src[0]); }
    for (int v = 0; v < 2; ++v) {
        double* t = tm[v];
        for (int a = 0; a < 5; ++a) for (int b = a + 1; b < 5; ++b) if (t[b] < t[a]) { double x = t[a]; t[a] = t[b]; t[b] = x; }
        printf("%-16s %-14s %8.0f %8.0f %8.0f   + 4 VALU + 1 KiB LDS reads per 32768 FLOP (the forward kernel's mix)\n", v ? "16x16x32 f16" : "32x32x16 f16", dn[0], t[0], t[2], t[4]);
    }
    // where the mix's power goes (16x16x32 f16, N(0,1) operands): LDS reads per 32768 FLOP x VALU on / off.  0.5 KiB = a wave keeping 64
    // query rows (each fragment feeds four MFMAs), 0.75 KiB = 48 rows
    struct M { const char* label; kfn f; } ms[] = {{"no LDS, no VALU", kmix<1, 0, 0>}, {"no LDS, 4 VALU", kmix<1, 0, 1>}, {"1 KiB LDS, no VALU", kmix<1, 4, 0>},
                                                  {"0.5 KiB LDS, 4 VALU", kmix<1, 2, 1>}, {"0.75 KiB LDS, 4 VALU", kmix<1, 3, 1>}, {"1 KiB LDS, 4 VALU", kmix<1, 4, 1>}};
    double tx[6][5];
    for (int r = 0; r < 5; ++r) for (int v = 0; v < 6; ++v) tx[v][r] = time_one(ms[v].f, d, src[0]);
    for (int v = 0; v < 6; ++v) {
        double* t = tx[v];
        for (int a = 0; a < 5; ++a) for (int b = a + 1; b < 5; ++b) if (t[b] < t[a]) { double x = t[a]; t[a] = t[b]; t[b] = x; }
        printf("%-16s %-14s %8.0f %8.0f %8.0f   mix: %s per 32768 FLOP\n", "16x16x32 f16", dn[0], t[0], t[2], t[4], ms[v].label);
    }
    return 0;
}
