// Dev lab: the f16x3 GEMM K-step (64x64 wave tile, hi/lo fragments re-read from LDS by ds_read_b128, three MFMAs per
// fragment pair) on RANDOM operands with v_mfma_f32_32x32x16_f16 (2 k16 groups x 12 MFMAs, 16 reads) versus
// v_mfma_f32_16x16x32_f16 (48 MFMAs, 16 reads): same FLOPs, same LDS bytes, same accumulator registers.  Reports
// TFLOP/s and the in-kernel clock (MI355X_MICROARCH.md "DVFS give-back" item 7 says the 16x16x32 shape holds a higher clock
// on random data).  1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// halves per LDS row ([32 hi | 32 lo] + pad), 128 A rows then 128 W rows; the pad that makes ds_read_b128 conflict-free
// depends on the fragment shape: 72 for 32-row fragments (lane -> row), 80 for 16-row fragments (lane -> row, k chunk)
template <int SHAPE> constexpr int row_halves() { return SHAPE == 32 ? 72 : 80; }
constexpr int ROWMAX = 80;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const _Float16* src, float* out, int iters, unsigned long long* clk) {
    extern __shared__ _Float16 lds[];
    constexpr int ROW = row_halves<SHAPE>();
    for (int i = threadIdx.x; i < 256 * ROW; i += 256) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    float s = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 32) {
        const int r = lane & 31, h = lane >> 5;
        const _Float16* A = lds + (wm * 64 + r) * ROW + 8 * h;
        const _Float16* W = lds + (128 + wn * 64 + r) * ROW + 8 * h;
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                f16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[t] = *reinterpret_cast<const f16x8*>(A + t * 32 * ROW + kg * 16);
                    al[t] = *reinterpret_cast<const f16x8*>(A + 32 + t * 32 * ROW + kg * 16);
                    wh[t] = *reinterpret_cast<const f16x8*>(W + t * 32 * ROW + kg * 16);
                    wl[t] = *reinterpret_cast<const f16x8*>(W + 32 + t * 32 * ROW + kg * 16);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ni], ah[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ni], al[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ni], ah[mi], acc[mi][ni], 0, 0, 0);
            }
        for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q >> 1][q & 1][i];
    } else {
        // 16x16x32: lane l holds row l & 15, k = 8 * (l >> 4) .. + 7 of a 16-row x 32-k fragment
        const int r = lane & 15, q = lane >> 4;
        const _Float16* A = lds + (wm * 64 + r) * ROW + 8 * q;
        const _Float16* W = lds + (128 + wn * 64 + r) * ROW + 8 * q;
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
            f16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(A + t * 16 * ROW);
                al[t] = *reinterpret_cast<const f16x8*>(A + 32 + t * 16 * ROW);
                wh[t] = *reinterpret_cast<const f16x8*>(W + t * 16 * ROW);
                wl[t] = *reinterpret_cast<const f16x8*>(W + 32 + t * 16 * ROW);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], ah[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], al[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], ah[mi], acc[mi][ni], 0, 0, 0);
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) s += acc[a][b][i];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    if (s == 1234.5678f) out[threadIdx.x] = s;
}

template <int SHAPE>
void run(const char* name, const _Float16* src, float* out, unsigned long long* clk, int wgs, int iters) {
    const size_t lds = 256 * ROWMAX * sizeof(_Float16);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a); hipLaunchKernelGGL(k<SHAPE>, dim3(wgs), dim3(256), lds, 0, src, out, iters, clk); (void)hipEventRecord(b);
        (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
    }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = double(wgs) * 4 * iters * 24 * 32768.0;  // per wave and iteration: 24 MFMAs of 32x32x16 worth
    printf("%-26s %4d WGs iters %6d: %8.3f ms  %7.0f TFLOP/s executed  clock %.3f GHz  %.1f cycles per 32x32x16-equivalent MFMA\n", name, wgs,
           iters, best, flops / (best * 1e-3) / 1e12, double(h[0]) / (double(h[1]) / 100.0) / 1e3, double(h[0]) / iters / 24.0 / (wgs > 256 ? 2 : 1));
}

int main() {
    float* out; unsigned long long* clk; _Float16* src;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&clk, 64); (void)hipMalloc(&src, 256 * ROWMAX * 2);
    std::vector<_Float16> hsrc(256 * ROWMAX);
    srand(1);
    for (auto& v : hsrc) v = _Float16((rand() / float(RAND_MAX) - 0.5f) * 4.0f);
    (void)hipMemcpy(src, hsrc.data(), hsrc.size() * 2, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r)
        for (int wgs : {256, 512}) {
            run<32>("32x32x16 (24 per step)", src, out, clk, wgs, 40000);
            run<16>("16x16x32 (48 per step)", src, out, clk, wgs, 40000);
        }
    return 0;
}
