// Dev harness (not product): times ablated variants of the attention kernel in ONE process
// (interleaved rounds), at the bench shape.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include "../pope_amd/csrc/attention_f32.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int AB>
float run(const float* qkv, float* out, int B, int N, int heads, size_t extra_lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_kernel<AB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        int(ATTN_LDS_BYTES + extra_lds));
    dim3 grid(unsigned((N + QB - 1) / QB) * heads * B);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(attn_f32_kernel<AB>, grid, dim3(256), ATTN_LDS_BYTES + extra_lds, 0, qkv, out, N, heads);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main(int argc, char** argv) {
    const int B = 64, N = 1531, heads = 6;
    const float scale = argc > 1 ? atof(argv[1]) : 1.0f;
    size_t n = size_t(B) * N * 3 * heads * 64;
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((s >> 8) * (1.0f / 8388608.0f) - 1.0f); }
    float *qkv, *out;
    hipMalloc(&qkv, n * 4); hipMalloc(&out, n / 3 * 4);
    hipMemcpy(qkv, h.data(), n * 4, hipMemcpyHostToDevice);
    const double gf = 4.0 * B * double(N) * N * heads * 64 / 1e9;
    for (int round = 0; round < 3; ++round) {
        printf("round %d (data scale %.2f)\n", round, scale);
        float t;
        t = run<0>(qkv, out, B, N, heads, 0);      printf("  full        3wg/CU : %.3f ms  %.1f TF\n", t, gf / t);
        t = run<0>(qkv, out, B, N, heads, 24000);  printf("  full        2wg/CU : %.3f ms  %.1f TF\n", t, gf / t);
        t = run<0>(qkv, out, B, N, heads, 60000);  printf("  full        1wg/CU : %.3f ms  %.1f TF\n", t, gf / t);
        t = run<1>(qkv, out, B, N, heads, 0);      printf("  no-softmax  3wg/CU : %.3f ms  %.1f TF(eq)\n", t, gf / t);
        t = run<1>(qkv, out, B, N, heads, 60000);  printf("  no-softmax  1wg/CU : %.3f ms  %.1f TF(eq)\n", t, gf / t);
        t = run<2>(qkv, out, B, N, heads, 0);      printf("  no-PV       3wg/CU : %.3f ms  (QK only: %.1f TF)\n", t, gf / 2 / t);
        t = run<6>(qkv, out, B, N, heads, 0);      printf("  no-QK no-PV 3wg/CU : %.3f ms\n", t);
        t = run<7>(qkv, out, B, N, heads, 0);      printf("  loads only  3wg/CU : %.3f ms\n", t);
    }
    return 0;
}
