// probe_isa.hip — pins the gfx950 lane layouts the attention kernels rely on, on real hardware:
//   (1) v_mfma_f32_32x32x16_f16 A/B/C fragment layout, (2) ds_read_b64_tr_b16 semantics,
//   (3) v_permlane32_swap semantics.  Prints PASS/FAIL per item; exit code = number of failures.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_isa.hip -o tools/probe_isa
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4v __attribute__((__vector_size__(4 * sizeof(short))));
#define LDS __attribute__((address_space(3)))

// C = A(32x16) * B(16x32) with A[i][k], B[k][n] given row-major in global memory
__global__ void k_mfma(const _Float16* A, const _Float16* B, float* C) {
    int l = threadIdx.x, hi = l >> 5, i = l & 31;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[i * 16 + 8 * hi + j]; b[j] = B[(8 * hi + j) * 32 + i]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * hi; C[row * 32 + i] = c[r]; }
}

// every lane supplies an arbitrary address (table), receives 4 elements
__global__ void k_tr(const int* lane_elem_off, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int l = threadIdx.x;
    i16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS i16x4v*)((LDS unsigned short*)lds + lane_elem_off[l]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

// C = A(16x32) * B(32x16): assumed lane layout of v_mfma_f32_16x16x32_f16 (fa_fwd_pp16.hip):
//   A: lane l holds row l & 15, k = 8 * (l >> 4) + j;  B: col l & 15, same k;  C: col l & 15, rows 4 * (l >> 4) + r, r = 0..3
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ void k_mfma16(const _Float16* A, const _Float16* B, float* C) {
    int l = threadIdx.x, g = l >> 4, i = l & 15;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[i * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + i]; }
    f32x4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];
}
__global__ void k_swap16(unsigned* out) {
    unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x * 2] = r[0];
    out[threadIdx.x * 2 + 1] = r[1];
}

__global__ void k_swap(unsigned* out) {
    unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x * 2] = r[0];
    out[threadIdx.x * 2 + 1] = r[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(99); } } while (0)

int main() {
    int fails = 0;
    {   // (1) MFMA layout, asymmetric random operands
        std::vector<_Float16> A(32 * 16), B(16 * 32);
        srand(1);
        for (auto& x : A) x = (_Float16)((rand() % 17 - 8) / 4.0f);
        for (auto& x : B) x = (_Float16)((rand() % 13 - 6) / 2.0f);
        _Float16 *dA, *dB; float* dC;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, 32 * 32 * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma<<<1, 64>>>(dA, dB, dC);
        std::vector<float> C(32 * 32);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
            double acc = 0; for (int k = 0; k < 16; ++k) acc += (double)A[i * 16 + k] * (double)B[k * 32 + n];
            maxerr = fmax(maxerr, fabs(acc - C[i * 32 + n]));
        }
        printf("[probe] mfma_f32_32x32x16_f16 A/B/C layout: %s (max err %.3g)\n", maxerr < 1e-3 ? "PASS" : "FAIL", maxerr);
        fails += !(maxerr < 1e-3);
    }
    {   // (2) ds_read_b64_tr_b16: result[16g + L][j] == lds[off[16g + 4j + (L>>2)] + (L&3)]
        std::vector<int> off(64);
        srand(7);
        for (int l = 0; l < 64; ++l) off[l] = 4 * (rand() % 1000);      // 8-byte aligned element offsets
        int* dOff; unsigned short* dOut;
        CK(hipMalloc(&dOff, 64 * 4)); CK(hipMalloc(&dOut, 64 * 4 * 2));
        CK(hipMemcpy(dOff, off.data(), 64 * 4, hipMemcpyHostToDevice));
        k_tr<<<1, 64>>>(dOff, dOut);
        std::vector<unsigned short> out(256);
        CK(hipMemcpy(out.data(), dOut, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            int g = l >> 4, L = l & 15;
            int expect = off[16 * g + 4 * j + (L >> 2)] + (L & 3);
            if (out[l * 4 + j] != expect) { if (bad < 8) printf("   tr mismatch lane %d j %d got %d expect %d\n", l, j, out[l * 4 + j], expect); ++bad; }
        }
        printf("[probe] ds_read_b64_tr_b16 semantics: %s\n", bad == 0 ? "PASS" : "FAIL");
        if (bad) {  // dump enough to reverse-engineer the real mapping
            for (int l = 0; l < 64; ++l) printf("   lane %2d off %4d -> %d %d %d %d\n", l, off[l], out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]);
        }
        fails += bad != 0;
    }
    {   // (3) permlane32_swap(a, b): r0 = {lanes<32: a own, lanes>=32: b of lane-32}; r1 = {lanes<32: a of lane+32, lanes>=32: b own}
        unsigned* dOut; CK(hipMalloc(&dOut, 128 * 4));
        k_swap<<<1, 64>>>(dOut);
        std::vector<unsigned> out(128);
        CK(hipMemcpy(out.data(), dOut, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            unsigned e0 = l < 32 ? 1000 + l : 2000 + (l - 32);
            unsigned e1 = l < 32 ? 1000 + (l + 32) : 2000 + l;
            if (out[2 * l] != e0 || out[2 * l + 1] != e1) { if (bad < 8) printf("   swap lane %d got (%u,%u) expect (%u,%u)\n", l, out[2*l], out[2*l+1], e0, e1); ++bad; }
        }
        printf("[probe] v_permlane32_swap semantics: %s\n", bad == 0 ? "PASS" : "FAIL");
        fails += bad != 0;
    }
    {   // (4) MFMA 16x16x32 layout
        std::vector<_Float16> A(16 * 32), B(32 * 16);
        srand(5);
        for (auto& x : A) x = (_Float16)((rand() % 17 - 8) / 4.0f);
        for (auto& x : B) x = (_Float16)((rand() % 13 - 6) / 2.0f);
        _Float16 *dA, *dB; float* dC;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, 16 * 16 * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma16<<<1, 64>>>(dA, dB, dC);
        std::vector<float> C(16 * 16);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
            double acc = 0; for (int k = 0; k < 32; ++k) acc += (double)A[i * 32 + k] * (double)B[k * 16 + n];
            maxerr = fmax(maxerr, fabs(acc - C[i * 16 + n]));
        }
        printf("[probe] mfma_f32_16x16x32_f16 A/B/C layout: %s (max err %.3g)\n", maxerr < 1e-3 ? "PASS" : "FAIL", maxerr);
        fails += !(maxerr < 1e-3);
    }
    {   // (5) permlane16_swap(a, b): expected r0 = {rows 0,2: own a; rows 1,3: b of lane-16}, r1 = {rows 0,2: a of lane+16; rows 1,3: own b}
        unsigned* dOut; CK(hipMalloc(&dOut, 128 * 4));
        k_swap16<<<1, 64>>>(dOut);
        std::vector<unsigned> out(128);
        CK(hipMemcpy(out.data(), dOut, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int odd = (l >> 4) & 1;
            unsigned e0 = !odd ? 1000 + l : 2000 + (l - 16);
            unsigned e1 = !odd ? 1000 + (l + 16) : 2000 + l;
            if (out[2 * l] != e0 || out[2 * l + 1] != e1) { if (bad < 8) printf("   swap16 lane %d got (%u,%u) expect (%u,%u)\n", l, out[2*l], out[2*l+1], e0, e1); ++bad; }
        }
        printf("[probe] v_permlane16_swap semantics: %s\n", bad == 0 ? "PASS" : "FAIL");
        fails += bad != 0;
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("[probe] device: %s arch %s CUs %d clock %d kHz mem %.1f GiB LDS/block %zu\n", prop.name, prop.gcnArchName,
           prop.multiProcessorCount, prop.clockRate, prop.totalGlobalMem / 1073741824.0, prop.sharedMemPerBlock);
    return fails;
}
