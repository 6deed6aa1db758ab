// fa_bwd_dkdv_common.hpp — workgroup shape and epilogue of the dK/dV kernel (fa_bwd.hip), kept apart from the tile loop so that
// alternative schedules of that loop can share them (round 3 measured a two-group ping-pong schedule against the lock-step one:
// profiles/r3_dkdv_ab.log).
#pragma once
#include "fa_device.hpp"
#include "fa_params.hpp"

namespace fa {

constexpr int kKvThreads = 512;
constexpr int kKvBlockN = 128;   // keys per workgroup (32 per key block, 4 key blocks)
constexpr int kKvBlockM = 64;    // query rows per staged tile (two 32-row halves)

// Wave w of a dK/dV workgroup owns key block kb = w & 3 (32 keys) and the 32-row half qh = w >> 2 of every Q / dO tile; its
// 2 x DB x 16 accumulator registers hold dK^T / dV^T partial sums of ITS q-half.  Epilogue, once the tile loop's last barrier has
// passed (all of LDS is scratch by then):
//   1) the qh = 1 waves hand their partial sums to their qh = 0 partner through LDS (fp32, [key block][register][lane], conflict-free);
//   2) the partner adds, applies the softmax scale to dK (flash_bwd_kernel.h:1652-1654), rounds and writes the staged 128-key tile;
//   3) all threads store whole rows.  dK and dV take turns in the same scratch.
// With a split query-head group (C ABI 3, p.n_split > 1) step 2 leaves the UNSCALED fp32 sum in this split's plane of the workspace
// instead, 16 bytes per lane per store, and fa_bwd_sum_splits_kernel finishes the job.
// The accumulators were last written by MFMAs issued from inline asm, which the hazard recogniser does not see: an 8-pass XDL
// write needs 11+ wait states before a VALU (v_accvgpr_read) may read it.  Pad explicitly, and tie every accumulator to a
// statement after the pad so no read can be scheduled above it.
template <typename T, int D, int SCRATCH_BYTES>
FA_DEV void dkdv_epilogue(const BwdKernelParams& p, FA_LDS char* smem, f32x16 (&dkacc)[D / 32], f32x16 (&dvacc)[D / 32],
                          int batch, int head_k, int split, int64_t k_row0, int n0, int keys_here, T* dk_base, T* dv_base) {
    constexpr int DB = D / 32, ROWB = D * 2, SLOTS = D / 8;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave & 3, qh = wave >> 2;
    const int key_row = kb * 32 + l31;
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int db = 0; db < DB; ++db) { asm volatile("" : "+a"(dkacc[db])); asm volatile("" : "+a"(dvacc[db])); }
    constexpr int XR = DB * 16;                                                // accumulator registers per lane and tensor
    FA_LDS float* xch = (FA_LDS float*)smem;                                   // 4 key blocks x XR x 64 lanes floats (64 KiB at d=128)
    FA_LDS char* out_t = smem + 4 * XR * 64 * 4;                               // staged output tile (32 KiB at d=128)
    static_assert(4 * XR * 64 * 4 + kKvBlockN * ROWB <= SCRATCH_BYTES, "epilogue scratch must fit the kernel's LDS");
    constexpr int O_CHUNKS = (kKvBlockN * SLOTS) / kKvThreads;
    const uint32_t dk_rowb = (uint32_t)(p.dk.row * 2), dv_rowb = (uint32_t)(p.dv.row * 2);
    const rsrc_t dk_rs = make_rsrc(dk_base, (uint32_t)(keys_here - 1) * dk_rowb + ROWB);
    const rsrc_t dv_rs = make_rsrc(dv_base, (uint32_t)(keys_here - 1) * dv_rowb + ROWB);
    auto reduce_and_store = [&](f32x16 (&acc)[DB], float mult, rsrc_t rs, uint32_t rowb) {
        if (qh == 1) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(kb * XR + db * 16 + r) * 64 + lane] = acc[db][r];
        }
        __syncthreads();
        if (qh == 0) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v4[e] = (acc[db][4 * g4 + e] + xch[(kb * XR + db * 16 + 4 * g4 + e) * 64 + lane]) * mult;
                    u32x2 w;
                    w.x = LP<T>::pack2(v4[0], v4[1]);
                    w.y = LP<T>::pack2(v4[2], v4[3]);
                    lds_write8(out_t, lds_tile_off<D>(key_row, 4 * db + g4) + 8 * hi, w);
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < O_CHUNKS; ++i) {
            const int chunk = tid + i * kKvThreads, row = chunk / SLOTS, slot = chunk % SLOTS;
            buf_store16(rs, (uint32_t)row * rowb + slot * 16, lds_read16(out_t, lds_tile_off<D>(row, slot)));
        }
        __syncthreads();                                                       // scratch is reused by the next tensor
    };
    if (p.n_split == 1) {
        reduce_and_store(dkacc, p.scale, dk_rs, dk_rowb);
        reduce_and_store(dvacc, 1.0f, dv_rs, dv_rowb);
        return;
    }
    // split group: a key's row is 4 * D bytes in the plane
    const int64_t plane = p.ws_rows * p.h_k * D;                                            // floats per (tensor, split)
    const int64_t row0 = (p.cu_seqlens_k != nullptr ? k_row0 : (int64_t)batch * p.seqlen_k) + n0;
    auto reduce_to_workspace = [&](f32x16 (&acc)[DB], int tensor) {
        if (qh == 1) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(kb * XR + db * 16 + r) * 64 + lane] = acc[db][r];
        }
        __syncthreads();
        if (qh == 0) {
            float* base = uniform_ptr(p.ws + ((int64_t)tensor * p.n_split + split) * plane + (row0 * p.h_k + head_k) * D);
            const rsrc_t rs = make_rsrc(base, (uint32_t)(keys_here - 1) * (uint32_t)(p.h_k * D * 4) + D * 4);
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    u32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w[e] = __builtin_bit_cast(uint32_t, acc[db][4 * g4 + e] + xch[(kb * XR + db * 16 + 4 * g4 + e) * 64 + lane]);
                    buf_store16(rs, (uint32_t)key_row * (uint32_t)(p.h_k * D * 4) + (32 * db + 8 * g4 + 4 * hi) * 4, w);   // rows >= keys_here fall outside the SRD
                }
        }
        __syncthreads();                                                       // scratch is reused by the next tensor
    };
    reduce_to_workspace(dkacc, 0);
    reduce_to_workspace(dvacc, 1);
}

}  // namespace fa
