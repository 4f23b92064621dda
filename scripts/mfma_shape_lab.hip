// Dev lab: bare f16 MFMA throughput, v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 (same FLOPs per wave and
// iteration, 4 / 16 independent accumulators), 1 and 2 waves per SIMD, plus the in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = _Float16(threadIdx.x * 1e-3f + i); y[i] = _Float16(blockIdx.x * 1e-4f - i); }
    float s = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 32) {
        f32x16 a[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 24; ++j) a[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a[j & 3], 0, 0, 0);
        for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += a[q][i];
    } else {
        f32x4 a[16] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 48; ++j) a[j & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a[j & 15], 0, 0, 0);
        for (int q = 0; q < 16; ++q) for (int i = 0; i < 4; ++i) s += a[q][i];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    if (s == 1234.5678f) out[threadIdx.x] = s;
}

template <int SHAPE>
void run(const char* name, float* out, unsigned long long* clk, int wgs) {
    const int iters = 20000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a); hipLaunchKernelGGL(k<SHAPE>, dim3(wgs), dim3(256), 0, 0, out, iters, clk); (void)hipEventRecord(b);
        (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = double(wgs) * 4 * iters * 24 * 32768.0;
    printf("%-28s %4d WGs: %8.3f ms  %7.0f TFLOP/s  in-kernel clock %.3f GHz\n", name, wgs, best, flops / (best * 1e-3) / 1e12,
           double(h[0]) / (double(h[1]) / 100.0) / 1e3);
}

int main() {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&clk, 64);
    for (int r = 0; r < 2; ++r)
        for (int wgs : {256, 512}) {
            run<32>("v_mfma_f32_32x32x16_f16", out, clk, wgs);
            run<16>("v_mfma_f32_16x16x32_f16", out, clk, wgs);
        }
    return 0;
}
