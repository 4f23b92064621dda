// Dev harness: how many independent VALU instructions hide under one v_mfma_f32_32x32x2_f32?
#include "../pope_amd/csrc/common.h"
#include <cstdio>

template <int NV, int KIND>  // KIND 0: v_fma  1: v_exp  2: v_pk_mul
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    f32x16 a0 = {}, a1 = {};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
    float v[8];
    f32x2 p[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = x + i;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = f32x2{x + i, y - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j & 1) a1 = mfma_32x32x2(y, x, a1); else a0 = mfma_32x32x2(x, y, a0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 7]) : "v"(x));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n & 7]));
                else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[n & 3]) : "v"(p[(n + 1) & 3]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    if (s == 1234.5678f) out[threadIdx.x] = s;
}

template <int NV, int KIND>
void run(float* out) {
    const int iters = 400;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, out, iters); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%s x%2d per MFMA: %.1f cycles per MFMA slot (at 2.4 GHz)\n", KIND == 0 ? "v_fma   " : KIND == 1 ? "v_exp   " : "v_pk_mul", NV,
           best * 1e-3 * 2.4e9 / (iters * 16.0));
}

int main() {
    float* out; hipMalloc(&out, 4096);
    run<0, 0>(out); run<0, 0>(out);
    run<1, 0>(out); run<2, 0>(out); run<4, 0>(out); run<6, 0>(out); run<8, 0>(out); run<10, 0>(out); run<12, 0>(out); run<16, 0>(out);
    run<1, 1>(out); run<2, 1>(out); run<4, 1>(out); run<6, 1>(out); run<8, 1>(out);
    run<1, 2>(out); run<2, 2>(out); run<4, 2>(out); run<8, 2>(out);
}
