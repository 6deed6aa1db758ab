// clockbench.hip -- chip-wide synthetic MFMA (+VALU, +LDS read) loops on MI355X: what a dependency-free, fully prefetched
// stream of a given composition sustains.  LDS reads are asm volatile with a delayed sink (never waited on per read: best case).
// Every variant is run 5x interleaved; check the loop bodies with llvm-objdump before quoting a number.
// Round 6: the mixed rows also exist for v_mfma_f32_16x16x32_f16 at the density the shipped head_dim-128 forward has NOW (k_mix16: per 4 MFMAs = 65536 FLOP
// two ds_read_b64_tr_b16 + one ds_read_b128 = 2 KiB, whose data ARE the MFMAs' A operands, and NV/5 VALU in the softmax's composition: fma, exp, cvt_pk, pk_max3),
// and as the kernel's own STRUCTURE without its dependencies (k_pp16: eight waves, two groups one phase apart, matrix phase = 68 MFMAs + 48 LDS reads at
// prefetch depth 2, softmax phase = 95 VALU, two barriers per tile, no LDS-DMA, no mask, no prologue / epilogue).
// Usage: clockbench [--rows SUBSTR[,SUBSTR..]] [--seconds S] [--reps N] [--grid N]   (--seconds: every timed launch lasts about S seconds; --grid 1: one CU, no power cap)
// = s_memtime ticks of one wave / hipEvent wall time.  Also reports achieved MFMA TFLOP/s of the
// pure-MFMA loop (the practical ceiling any attention kernel is chasing on this box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

// ---- round 6: right-shape probes -------------------------------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) { if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); } }
static __device__ __forceinline__ void MF16(f32x4& acc, const u32x4& a, const f16x8& b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
// one softmax-like VALU instruction, kind K (fma, exp, fma, exp, cvt_pk, pk_max3) on register x (y as a second source)
template <int K> static __device__ __forceinline__ void vk(float& x, const float& y) {
    if constexpr (K % 6 == 0 || K % 6 == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    else if constexpr (K % 6 == 1 || K % 6 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if constexpr (K % 6 == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    else asm volatile("v_pk_maximum3_f16 %0, %0, %1, %1" : "+v"(x) : "v"(y));
}
#define VK(k, x, y) vk<k>(x, y)
// fragment reads with immediate offsets (no address arithmetic in the loops); never waited on here: the caller counts lgkmcnt
template <int OFF> static __device__ __forceinline__ void rd_tr(u32x4& dst, const unsigned int& base) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=v"(lo), "=v"(hi) : "v"(base), "n"(OFF), "n"(OFF + 4096));
    dst = u32x4{lo.x, lo.y, hi.x, hi.y};
}
template <int OFF> static __device__ __forceinline__ void rd_b128(u32x4& dst, const unsigned int& base) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(OFF + 2048));
}
template <int N> static __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
static __device__ __forceinline__ void fill_lds(unsigned int* lds, const _Float16* rnd) {      // well-formed fp16 pairs everywhere (no NaN / Inf patterns: they would stop toggling)
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        const _Float16 x = rnd[(i * 2) & 4095], y = rnd[(i * 2 + 1) & 4095];
        lds[i] = (unsigned int)__builtin_bit_cast(unsigned short, x) | ((unsigned int)__builtin_bit_cast(unsigned short, y) << 16);
    }
    __syncthreads();
}
// NV10 = VALU instructions per 10 MFMA pairs (31: the shipped head_dim-128 forward, 3.1 per 32768 FLOP; 40: the round-1..3 kernel; 52: head_dim 64; 0: none).  LDS: 0 none, 1 = 1 KiB per pair
template <int NV10, int LDS>
__global__ __launch_bounds__(512, 2) void k_mix16(unsigned long long* out, const _Float16* rnd, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[16384];
    fill_lds(lds, rnd);
    const unsigned int lbase = (unsigned int)(size_t)(lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 256);      // b128: lane-linear 1 KiB per wave
    const unsigned int tbase = (unsigned int)(size_t)(lds + (threadIdx.x & 63) * 2 + (threadIdx.x >> 6) * 256);      // tr_b64: 8 bytes per lane
    f16x8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = rnd[(threadIdx.x * 8 + i) & 4095]; b1[i] = rnd[(threadIdx.x * 8 + i + 77) & 4095]; }
    u32x4 fa[2], fb[2];
    for (int i = 0; i < 2; ++i) { fa[i] = __builtin_bit_cast(u32x4, b0); fb[i] = __builtin_bit_cast(u32x4, b1); }
    f32x4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float r[12];
    for (int i = 0; i < 12; ++i) r[i] = (float)b0[i & 7] * 0.25f;
    const float half = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        static_for<0, 10>([&](auto gc) {          // 10 groups = 20 pairs = 40 MFMAs per trip
            constexpr int g = decltype(gc)::value, cur = g & 1, nxt = cur ^ 1;
            if constexpr (LDS != 0) {
                constexpr int off = (g & 7) * 8192;      // (immediate offsets: no address arithmetic in the loop)
                rd_tr<off>(fa[nxt], tbase);
                rd_b128<off>(fb[nxt], lbase);
                wait_lgkm<3>();      // the previous group's three reads have landed (in-order return)
            }
            constexpr int nv = (NV10 * 2) / 10 + (g < (NV10 * 2) % 10 ? 1 : 0);      // VALU of this group (2 pairs)
            MF16(acc[(4 * g) % 24], fa[cur], b0); MF16(acc[(4 * g + 1) % 24], fa[cur], b1);
            static_for<0, (nv + 1) / 2>([&](auto kc) { constexpr int k = decltype(kc)::value; VK(k, r[(g + k) % 12], half); });
            MF16(acc[(4 * g + 2) % 24], fb[cur], b0); MF16(acc[(4 * g + 3) % 24], fb[cur], b1);
            static_for<(nv + 1) / 2, nv>([&](auto kc) { constexpr int k = decltype(kc)::value; VK(k, r[(g + k) % 12], half); });
        });
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0;
    for (int i = 0; i < 24; ++i) { asm volatile("" : "+v"(acc[i])); s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
    for (int i = 0; i < 12; ++i) s += r[i];
    if (s == 1234.5f) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}
// The forward kernel's two-group ping-pong without its dependencies: per trip and wave one matrix phase (NPAIR fragment steps of two MFMAs: the first half
// reads two ds_read_b64_tr_b16 per fragment, the second half one ds_read_b128, requested two fragments ahead, plus two row-sum pairs = 68 MFMAs at head_dim 128)
// and one softmax phase (NVALU instructions), separated by workgroup barriers; waves 4-7 run one phase behind waves 0-3.
template <int NPAIR, int NVALU, int WITH_LDS>
__global__ __launch_bounds__(512, 2) void k_pp16(unsigned long long* out, const _Float16* rnd, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[16384];
    fill_lds(lds, rnd);
    const unsigned int lbase = (unsigned int)(size_t)(lds + (threadIdx.x & 63) * 4 + ((threadIdx.x >> 6) & 3) * 256);
    const unsigned int tbase = (unsigned int)(size_t)(lds + (threadIdx.x & 63) * 2 + ((threadIdx.x >> 6) & 3) * 256);
    const int group = threadIdx.x >> 8;
    f16x8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = rnd[(threadIdx.x * 8 + i) & 4095]; b1[i] = rnd[(threadIdx.x * 8 + i + 77) & 4095]; }
    u32x4 fr[3];
    for (int i = 0; i < 3; ++i) fr[i] = __builtin_bit_cast(u32x4, b0);
    f32x4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = (float)b0[i & 7] * 0.25f;
    const float half = 0.5f;
    auto rd = [&](auto jc, u32x4& dst) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, off = (j & 7) * 8192;
        if constexpr (WITH_LDS != 0) {
            if constexpr (j < NPAIR / 2) rd_tr<off>(dst, tbase);
            else rd_b128<off>(dst, lbase);
        }
    };
    auto m_phase = [&]() __attribute__((always_inline)) {
        rd(IC<0>{}, fr[0]); rd(IC<1>{}, fr[1]);
        static_for<0, NPAIR>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 2 < NPAIR) rd(IC<j + 2>{}, fr[(j + 2) % 3]);
            if constexpr (WITH_LDS != 0) {      // fragment j has landed: the reads behind it are those of j + 1 and j + 2 (two instructions each in the first half)
                constexpr int behind = (j + 1 < NPAIR ? (j + 1 < NPAIR / 2 ? 2 : 1) : 0) + (j + 2 < NPAIR ? (j + 2 < NPAIR / 2 ? 2 : 1) : 0);
                wait_lgkm<behind>();
            }
            MF16(acc[(2 * j) % 24], fr[j % 3], b0); MF16(acc[(2 * j + 1) % 24], fr[j % 3], b1);
        });
        MF16(acc[0], fr[0], b0); MF16(acc[1], fr[0], b1); MF16(acc[2], fr[1], b0); MF16(acc[3], fr[1], b1);      // the four row-sum MFMAs
    };
    auto s_phase = [&]() __attribute__((always_inline)) {
        static_for<0, NVALU>([&](auto kc) { constexpr int k = decltype(kc)::value; VK(k, r[k % 16], half); });
    };
    if (group == 1) __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        m_phase();
        __builtin_amdgcn_s_barrier();
        s_phase();
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (group == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0;
    for (int i = 0; i < 24; ++i) { asm volatile("" : "+v"(acc[i])); s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
    for (int i = 0; i < 16; ++i) s += r[i];
    if (s == 1234.5f) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}
struct Variant { const char* label; void (*fn)(unsigned long long*, const _Float16*, int); int wps; double flop_per_iter_wave; double tf[16]; double cyc[16]; int n; };
static int g_grid = 256;
static double g_seconds = 0.0;
// one timed launch; `iters` sized from a calibration launch when --seconds is given.  Also returns wave 0's s_memtime ticks per iteration (k_mix16 / k_pp16 only).
static double time_one(Variant& v, unsigned long long* d, _Float16* rnd, double* ticks_per_iter) {
    const int threads = 256 * v.wps;
    int iters = (int)(20000.0 * 524288.0 / v.flop_per_iter_wave);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(v.fn, dim3(g_grid), dim3(threads), 0, 0, d, rnd, 100); CK(hipDeviceSynchronize());
    float ms;
    if (g_seconds > 0.0) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(v.fn, dim3(g_grid), dim3(threads), 0, 0, d, rnd, iters / 8); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        iters = (int)((iters / 8) * (g_seconds * 1e3 / ms));
    }
    CK(hipMemset(d, 0, 16));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(v.fn, dim3(g_grid), dim3(threads), 0, 0, d, rnd, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    *ticks_per_iter = (double)h[0] / iters;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (double)g_grid * (threads / 64) * iters * v.flop_per_iter_wave / (ms * 1e9);
}
static bool wanted(const char* label, const char* rows) {
    if (!rows) return true;
    char buf[512]; strncpy(buf, rows, sizeof(buf) - 1); buf[sizeof(buf) - 1] = 0;
    for (char* t = strtok(buf, ","); t; t = strtok(nullptr, ",")) if (strstr(label, t)) return true;
    return false;
}
int main(int argc, char** argv) {
    const char* rows = nullptr; int REP = 5;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--rows") && i + 1 < argc) rows = argv[++i];
        else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) g_seconds = atof(argv[++i]);
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) REP = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--grid") && i + 1 < argc) g_grid = atoi(argv[++i]);
        else { printf("usage: clockbench [--rows SUBSTR[,..]] [--seconds S] [--reps N<=16] [--grid N]\n"); return 2; }
    }
    if (REP < 1) REP = 1; if (REP > 16) REP = 16;
    unsigned long long* d; CK(hipMalloc(&d, 64));
    _Float16 hr[4096]; srand(3);
    for (auto& x : hr) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    _Float16* rnd; CK(hipMalloc(&rnd, sizeof(hr))); CK(hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice));
    const double F32 = 16.0 * 2 * 32 * 32 * 16, M16 = 2.0 * 16 * 16 * 32;
    Variant all[] = {
        {"MFMA only, 1 wave/SIMD", k_mfma<0, 0, 0>, 1, F32},
        {"MFMA only, 2 waves/SIMD", k_mfma<0, 0, 0>, 2, F32},
        {"16x16x32 MFMA only, 2 waves/SIMD", k_mfma16, 2, F32},
        {"MFMA + 4 VALU, 1 wave/SIMD", k_mfma<1, 0, 0>, 1, F32},
        {"MFMA + 4 VALU, 2 waves/SIMD", k_mfma<1, 0, 0>, 2, F32},
        {"MFMA + 0.5 KB LDS/MFMA (b128), 2 w/SIMD", k_mfma<0, 2, 0>, 2, F32},
        {"MFMA + 1 KB LDS/MFMA (b128), 2 w/SIMD", k_mfma<0, 4, 0>, 2, F32},
        {"MFMA + 4 VALU + 0.5 KB LDS (b128), 2 w/SIMD", k_mfma<1, 2, 0>, 2, F32},
        {"MFMA + 4 VALU + 1 KB LDS (b128), 2 w/SIMD", k_mfma<1, 4, 0>, 2, F32},
        {"MFMA + 4 VALU + 1 KB LDS (b128+tr mix), 2 w/SIMD", k_mfma<1, 4, 1>, 2, F32},
        {"MFMA + 4 VALU + 0.5 KB LDS (b128), 1 w/SIMD", k_mfma<1, 2, 0>, 1, F32},
        // round 6: v_mfma_f32_16x16x32_f16, A operands from the LDS reads, VALU per 32768 FLOP as named, 1 KiB of LDS reads (b128 + tr mix) per 32768 FLOP
        {"16x16x32 mix: operands from LDS, no VALU, 2 w/SIMD", k_mix16<0, 1>, 2, 40 * M16},
        {"16x16x32 mix: 3.1 VALU, no LDS, 2 w/SIMD", k_mix16<31, 0>, 2, 40 * M16},
        {"16x16x32 mix: 3.1 VALU + 1 KiB LDS (fwd d128 now), 2 w/SIMD", k_mix16<31, 1>, 2, 40 * M16},
        {"16x16x32 mix: 4.0 VALU + 1 KiB LDS (fwd d128 r1-r3), 2 w/SIMD", k_mix16<40, 1>, 2, 40 * M16},
        {"16x16x32 mix: 5.2 VALU + 1 KiB LDS (fwd d64), 2 w/SIMD", k_mix16<52, 1>, 2, 40 * M16},
        {"16x16x32 mix: 3.1 VALU + 1 KiB LDS, 1 w/SIMD", k_mix16<31, 1>, 1, 40 * M16},
        {"16x16x32 ping-pong d128: 68 MFMA + 48 LDS | 95 VALU, 8 waves", k_pp16<32, 95, 1>, 2, 68 * M16},
        {"16x16x32 ping-pong d128: 68 MFMA | 95 VALU, no LDS reads", k_pp16<32, 95, 0>, 2, 68 * M16},
        {"16x16x32 ping-pong d128: 68 MFMA + 48 LDS | no VALU", k_pp16<32, 0, 1>, 2, 68 * M16},
        {"16x16x32 ping-pong d64: 68 MFMA + 48 LDS | 190 VALU, 8 waves", k_pp16<32, 190, 1>, 2, 68 * M16},
    };
    Variant* vs[64]; int NV = 0;
    for (auto& v : all) if (wanted(v.label, rows)) { v.n = 0; vs[NV++] = &v; }
    for (int r = 0; r < REP; ++r)             // interleaved: every variant once per round, so drift hits all of them alike
        for (int i = 0; i < NV; ++i) { double tk; vs[i]->tf[vs[i]->n] = time_one(*vs[i], d, rnd, &tk); vs[i]->cyc[vs[i]->n++] = tk; }
    printf("%-64s %8s %8s %8s  (TFLOP/s over %d interleaved runs%s; grid %d)\n", "variant", "min", "median", "max", REP, g_seconds > 0 ? ", timed launches of --seconds each" : "", g_grid);
    for (int i = 0; i < NV; ++i) {
        Variant& v = *vs[i];
        for (int a = 0; a < v.n; ++a) for (int b = a + 1; b < v.n; ++b) if (v.tf[b] < v.tf[a]) { double t = v.tf[a]; v.tf[a] = v.tf[b]; v.tf[b] = t; t = v.cyc[a]; v.cyc[a] = v.cyc[b]; v.cyc[b] = t; }
        printf("%-64s %8.0f %8.0f %8.0f\n", v.label, v.tf[0], v.tf[v.n / 2], v.tf[v.n - 1]);
    }
    printf("# wave-0 s_memtime ticks per loop trip (median run):\n");
    for (int i = 0; i < NV; ++i) if (vs[i]->cyc[vs[i]->n / 2] > 0) printf("# ticks  %-64s %10.3f\n", vs[i]->label, vs[i]->cyc[vs[i]->n / 2]);
    return 0;
}
