// Shared 128x128x32 fp32-MFMA tile mainloop (see gemm_f32.hip for the design notes).
#pragma once
#include "common.h"

namespace gemm_core {

constexpr int BM = 128, BN = 128, BK = 32, LDS_ST = 36;
constexpr int THREADS = 256;
constexpr size_t LDS_BYTES = size_t(2) * (BM + BN) * LDS_ST * sizeof(float);

// acc[mi][ni] += A_tile . W_tile^T over K.  `la(row, k)` / `lw(row, k)` return the four
// K-contiguous floats at (tile row `row` in [0,128), absolute k) or zeros outside the operand.
// Register-staged double buffering: the global loads of K-tile t+1 are issued before the 64
// MFMAs of K-tile t and written to the other LDS buffer after them; one barrier per K-tile.
template <class LoadA, class LoadW>
__device__ __forceinline__ void mainloop(LoadA la, LoadW lw, int K, float* smem, f32x16 (&acc)[2][2]) {
    float* As = smem;                    // [2][BM][LDS_ST]
    float* Ws = smem + 2 * BM * LDS_ST;  // [2][BN][LDS_ST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;

#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    f32x4 ra[4], rw[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = la(srow + 32 * i, k0 + scol);
            rw[i] = lw(srow + 32 * i, k0 + scol);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&As[(buf * BM + srow + 32 * i) * LDS_ST + scol]) = ra[i];
            *reinterpret_cast<f32x4*>(&Ws[(buf * BN + srow + 32 * i) * LDS_ST + scol]) = rw[i];
        }
    };

    const int nk = (K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const float* a_base = &As[(cur * BM + wm * 64 + r) * LDS_ST + 4 * h];
        const float* w_base = &Ws[(cur * BN + wn * 64 + r) * LDS_ST + 4 * h];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const f32x4*>(a_base + t * 32 * LDS_ST + 8 * j);
                b[t] = *reinterpret_cast<const f32x4*>(w_base + t * 32 * LDS_ST + 8 * j);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = mfma_32x32x2(a[mi][s], b[ni][s], acc[mi][ni]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }
}

// Visit every accumulator element of this lane: f(row_in_tile, col_in_tile, value).
template <class F>
__device__ __forceinline__ void for_each_output(const f32x16 (&acc)[2][2], F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                f(wm * 64 + mi * 32 + mfma32_row(i, h), wn * 64 + ni * 32 + r, acc[mi][ni][i]);
}

}  // namespace gemm_core
