// atomicbench.hip -- can dQ be accumulated across key blocks with fp32 atomics (the reference's own backward, flash_bwd_kernel.h: dq_accum +
// atomicAdd) at the rate a single fused backward kernel would need on an MI355X?  (profiles/NOTEBOOK.md 7, item 0.)
//
// The access pattern of that kernel at BASELINE configs[3] (b4 h32 s8192 d128), nothing else: one workgroup per (head, 128-key block) =
// 128 heads x 64 blocks; each sweeps the 128 query tiles (64 rows x 128 d fp32 = 32 KiB) of its head's 4 MiB dq_accum and adds a tile's
// worth of values per step: 16 atomics of 4 B per thread and tile, lane-contiguous.  34.4 GB of adds per backward in total.
//   mode 0: global_atomic_add_f32 (no return), workgroups of one head spread over the XCDs as the hardware deals them (blockIdx % 8)
//   mode 1: the same, all 64 key blocks of a head on ONE XCD (head = f(blockIdx % 8, ...)): the head's 4 MiB stay in that XCD's L2
//   mode 2: plain stores of the same bytes (what the traffic alone costs), mapping of mode 1
//   mode 3: mode 1 with the tile order staggered by key block (block j starts at tile j): neighbours do not chase each other
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int HEADS = 128, KBLOCKS = 64, QTILES = 128, TILE_FLOATS = 64 * 128;
template <int MODE>
__global__ __launch_bounds__(512) void k(float* acc, float v) {
    int head, kb;
    if (MODE == 0) { head = blockIdx.x / KBLOCKS; kb = blockIdx.x % KBLOCKS; }
    else { const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8; head = (slot / KBLOCKS) * 8 + xcd; kb = slot % KBLOCKS; }      // 16 heads per XCD
    float* base = acc + (size_t)head * QTILES * TILE_FLOATS;
    for (int t = 0; t < QTILES; ++t) {
        const int tile = MODE == 3 ? (t + 2 * kb) % QTILES : t;
        float* p = base + (size_t)tile * TILE_FLOATS + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 2) __builtin_nontemporal_store(v + i, p + i * 512);
            else __hip_atomic_fetch_add(p + i * 512, v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <int MODE>
static double run(float* d) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(HEADS * KBLOCKS), dim3(512), 0, 0, d, 1.0f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(HEADS * KBLOCKS), dim3(512), 0, 0, d, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
int main() {
    float* d; const size_t bytes = (size_t)HEADS * QTILES * TILE_FLOATS * 4;
    CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
    const double gb = (double)HEADS * KBLOCKS * QTILES * TILE_FLOATS * 4 / 1e9;
    const char* names[4] = {"atomic add, heads spread over XCDs", "atomic add, a head's key blocks on one XCD", "plain stores, one XCD per head", "atomic add, one XCD per head, staggered tiles"};
    double ms[4][3];
    for (int r = 0; r < 3; ++r) { ms[0][r] = run<0>(d); ms[1][r] = run<1>(d); ms[2][r] = run<2>(d); ms[3][r] = run<3>(d); }
    printf("dq_accum %.0f MiB, %.1f GB of fp32 adds per pass (b4 h32 s8192 d128 backward, 128-key blocks)\n", bytes / 1048576.0, gb);
    for (int m = 0; m < 4; ++m) {
        double best = ms[m][0]; for (int r = 1; r < 3; ++r) if (ms[m][r] < best) best = ms[m][r];
        printf("%-48s %8.2f ms  %7.0f GB/s\n", names[m], best, gb / best * 1e3);
    }
    float h[4]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost)); printf("(check: acc[0] = %.0f)\n", h[0]);
    return 0;
}
