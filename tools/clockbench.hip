// clockbench.hip -- chip-wide synthetic MFMA (+VALU, +LDS read) loops on MI355X: what a dependency-free, fully prefetched
// stream of a given composition sustains.  LDS reads are asm volatile with a delayed sink (never waited on per read: best case).
// Every variant is run 5x interleaved; check the loop bodies with llvm-objdump before quoting a number.
// = s_memtime ticks of one wave / hipEvent wall time.  Also reports achieved MFMA TFLOP/s of the
// pure-MFMA loop (the practical ceiling any attention kernel is chasing on this box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define VALU4(a, b, c, d) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int WITH_VALU, int LDS_PER_4 = 0, int TR_MIX = 0>
__global__ __launch_bounds__(512, 2) void k_mfma(unsigned long long* out, const _Float16* rnd, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const unsigned int lbase = (unsigned int)(size_t)(lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 256);
    u32x4 ld[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};  // independent destinations: no serial chain through the reads
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
            for (int q = 0; q < LDS_PER_4; ++q) {
                // sink the tuple written 4 reads ago (forces its wait late, like a prefetched fragment), then overwrite it
                asm volatile("" :: "v"(ld[q & 3]));
                const unsigned int a0 = lbase + (((it * 4 + j + q * 7) & 7) * 8192u);
                if (TR_MIX && (q & 1)) {   // kernel-like mix: a pair of ds_read_b64_tr_b16 (2 x 512 B) in place of one b128
                    unsigned long long t0_, t1_;
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096" : "=v"(t0_), "=v"(t1_) : "v"(a0));
                    ld[q & 3] = u32x4{(unsigned int)t0_, (unsigned int)(t0_ >> 32), (unsigned int)t1_, (unsigned int)(t1_ >> 32)};
                } else {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(ld[q & 3]) : "v"(a0));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 8; ++i) s += r[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 lacc = ld[0] ^ ld[1] ^ ld[2] ^ ld[3];
    if (s == 1234.5f || lacc.x + lacc.y + lacc.z + lacc.w == 77u) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
// the same FLOPs per iteration from v_mfma_f32_16x16x32_f16 (32 per iteration, 8 accumulators): the shape the head_dim-128 16x16x32 kernels use
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void k_mfma16(unsigned long long* out, const _Float16* rnd, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = rnd[(threadIdx.x * 8 + i) & 4095]; b[i] = rnd[(threadIdx.x * 8 + i + 77) & 4095]; }
    f32x4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += d[i][0] + d[i][1] + d[i][2] + d[i][3];
    if (s == 1234.5f) out[1] = 1;
}
struct Variant { const char* label; void (*fn)(unsigned long long*, const _Float16*, int); int wps; double tf[8]; int n; };
static double time_one(void (*k)(unsigned long long*, const _Float16*, int), unsigned long long* d, _Float16* rnd, int wps) {
    const int iters = 20000, threads = 256 * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, rnd, 100); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, rnd, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 256.0 * (threads / 64) * iters * 16.0 * 2 * 32 * 32 * 16 / (ms * 1e9);
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    _Float16 hr[4096]; srand(3);
    for (auto& x : hr) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    _Float16* rnd; CK(hipMalloc(&rnd, sizeof(hr))); CK(hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice));
    Variant vs[] = {
        {"MFMA only, 1 wave/SIMD", k_mfma<0, 0, 0>, 1},
        {"MFMA only, 2 waves/SIMD", k_mfma<0, 0, 0>, 2},
        {"16x16x32 MFMA only, 2 waves/SIMD", k_mfma16, 2},
        {"MFMA + 4 VALU, 1 wave/SIMD", k_mfma<1, 0, 0>, 1},
        {"MFMA + 4 VALU, 2 waves/SIMD", k_mfma<1, 0, 0>, 2},
        {"MFMA + 0.5 KB LDS/MFMA (b128), 2 w/SIMD", k_mfma<0, 2, 0>, 2},
        {"MFMA + 1 KB LDS/MFMA (b128), 2 w/SIMD", k_mfma<0, 4, 0>, 2},
        {"MFMA + 4 VALU + 0.5 KB LDS (b128), 2 w/SIMD", k_mfma<1, 2, 0>, 2},
        {"MFMA + 4 VALU + 1 KB LDS (b128), 2 w/SIMD", k_mfma<1, 4, 0>, 2},
        {"MFMA + 4 VALU + 1 KB LDS (b128+tr mix), 2 w/SIMD", k_mfma<1, 4, 1>, 2},
        {"MFMA + 4 VALU + 0.5 KB LDS (b128), 1 w/SIMD", k_mfma<1, 2, 0>, 1},
    };
    const int NV = sizeof(vs) / sizeof(vs[0]), REP = 5;
    for (auto& v : vs) v.n = 0;
    for (int r = 0; r < REP; ++r)             // interleaved: every variant once per round, so drift hits all of them alike
        for (int i = 0; i < NV; ++i) vs[i].tf[vs[i].n++] = time_one(vs[i].fn, d, rnd, vs[i].wps);
    printf("%-52s %8s %8s %8s  (TFLOP/s over %d interleaved runs)\n", "variant", "min", "median", "max", REP);
    for (auto& v : vs) {
        for (int a = 0; a < v.n; ++a) for (int b = a + 1; b < v.n; ++b) if (v.tf[b] < v.tf[a]) { double t = v.tf[a]; v.tf[a] = v.tf[b]; v.tf[b] = t; }
        printf("%-52s %8.0f %8.0f %8.0f\n", v.label, v.tf[0], v.tf[v.n / 2], v.tf[v.n - 1]);
    }
    return 0;
}
