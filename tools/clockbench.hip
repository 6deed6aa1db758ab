// clockbench.hip — sustained shader clock of MI355X under a chip-wide MFMA load (random operands),
// = s_memtime ticks of one wave / hipEvent wall time.  Also reports achieved MFMA TFLOP/s of the
// pure-MFMA loop (the practical ceiling any attention kernel is chasing on this box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define VALU4(a, b, c, d) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int WITH_VALU, int LDS_PER_4 = 0>
__global__ __launch_bounds__(512, 2) void k_mfma(unsigned long long* out, const _Float16* rnd, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const unsigned int* lp = lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 256;
    u32x4 lacc = {0, 0, 0, 0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = rnd[(threadIdx.x * 8 + i) & 4095]; b[i] = rnd[(threadIdx.x * 8 + i + 77) & 4095]; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = (float)a[i];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            if (WITH_VALU) VALU4(r[0], r[1], r[2], r[3]);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            if (WITH_VALU) VALU4(r[4], r[5], r[6], r[7]);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            if (WITH_VALU) VALU4(r[0], r[1], r[2], r[3]);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            if (WITH_VALU) VALU4(r[4], r[5], r[6], r[7]);
#pragma unroll
            for (int q = 0; q < LDS_PER_4; ++q) { u32x4 t = *(const u32x4*)(lp + ((it * 4 + j + q * 7) & 7) * 2048); lacc ^= t; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 1234.5f || lacc.x + lacc.y + lacc.z + lacc.w == 77u) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int V, int L = 0>
void run(unsigned long long* d, _Float16* rnd, int waves_per_simd, const char* label) {
    const int iters = 20000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_mfma<V, L><<<256, threads>>>(d, rnd, 100); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_mfma<V, L><<<256, threads>>>(d, rnd, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    const double flops = 256.0 * (threads / 64) * iters * 16.0 * 2 * 32 * 32 * 16;
    printf("%-40s %7.3f ms  s_memtime %.4g ticks -> %.3f GHz ; %.0f TFLOP/s ; %.1f ticks/MFMA/SIMD\n", label, ms, (double)h, h / (ms * 1e6), flops / (ms * 1e9),
           (double)h / (iters * 16.0 * waves_per_simd));
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    _Float16 hr[4096]; srand(3);
    for (auto& x : hr) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    _Float16* rnd; CK(hipMalloc(&rnd, sizeof(hr))); CK(hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice));
    run<0>(d, rnd, 1, "MFMA only, 1 wave/SIMD, all CUs");
    run<0>(d, rnd, 2, "MFMA only, 2 waves/SIMD, all CUs");
    run<1>(d, rnd, 1, "MFMA + 4 VALU, 1 wave/SIMD, all CUs");
    run<1>(d, rnd, 2, "MFMA + 4 VALU, 2 waves/SIMD, all CUs");
    run<0, 4>(d, rnd, 2, "MFMA + 1 LDS b128/MFMA, 2 waves/SIMD");
    run<0, 8>(d, rnd, 2, "MFMA + 2 LDS b128/MFMA, 2 waves/SIMD");
    run<1, 6>(d, rnd, 2, "MFMA + 4 VALU + 1.5 LDS b128, 2 waves/SIMD");
    run<1, 4>(d, rnd, 2, "MFMA + 4 VALU + 1 LDS b128, 2 waves/SIMD");
    run<0>(d, rnd, 2, "MFMA only, 2 waves/SIMD (repeat)");
    return 0;
}
