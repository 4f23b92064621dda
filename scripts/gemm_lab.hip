// Dev harness (not product): ablations of the fp32 GEMM tile kernel, one process, interleaved.
#include "../pope_amd/csrc/gemm_f32.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

struct P { const float* A; const float* W; float* C; int M, N, K; };

template <int LAB, int EPI>  // EPI 0: none (keeps acc live), 1: coalesced store
__global__ __launch_bounds__(THREADS, 3) void lab_kernel(P g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    f32x16 acc[2][2];
    if constexpr (LAB & 8) {
        mainloop<(LAB & 3)>(
            fn_loader([&](int row, int k) { f32x4 v = {0, 0, 0, 0}; if (m0 + row < g.M && k < g.K) v = *reinterpret_cast<const f32x4*>(g.A + size_t(m0 + row) * g.K + k); return v; }),
            fn_loader([&](int row, int k) { f32x4 v = {0, 0, 0, 0}; if (n0 + row < g.N && k < g.K) v = *reinterpret_cast<const f32x4*>(g.W + size_t(n0 + row) * g.K + k); return v; }),
            g.K, smem, acc);
    } else {
        mainloop<(LAB & 3)>(BufferLoader(g.A, g.M, g.K, m0), BufferLoader(g.W, g.N, g.K, n0), g.K, smem, acc);
    }
    if constexpr (EPI == 1) {
        epilogue_rows(acc, smem, [&](int tr, int tc, f32x4 v) {
            const int row = m0 + tr, col = n0 + tc;
            if (row < g.M && col < g.N) *reinterpret_cast<f32x4*>(g.C + size_t(row) * g.N + col) = v;
        });
    } else {
        float s = 0;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) s += acc[a][b][i];
        if (s == 1234.5678f) g.C[threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256, 3) void mfma_only(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        a0 = mfma_32x32x2(x, y, a0); a1 = mfma_32x32x2(y, x, a1);
        a2 = mfma_32x32x2(x, x, a2); a3 = mfma_32x32x2(y, y, a3);
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 1234.5678f) out[threadIdx.x] = s;
}

template <int LAB, int EPI>
float run(P g, size_t extra) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(lab_kernel<LAB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_BYTES + extra));
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((lab_kernel<LAB, EPI>), dim3(tiles), dim3(THREADS), LDS_BYTES + extra, 0, g);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

float run_persistent(P p, int wg_per_cu) {
    GemmParams g = {};
    g.A = p.A; g.W = p.W; g.C = p.C; g.M = p.M; g.N = p.N; g.K = p.K; g.lda = p.K; g.ldw = p.K; g.ldc = p.N;
    const int n_tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((gemm_nt_f32_persistent_kernel<EPI_BIAS>), dim3(wg_per_cu * 256), dim3(THREADS), LDS_BYTES, 0, g, n_tiles);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    const int M = 64 * 1531;
    size_t na = size_t(M) * 1536, nw = size_t(1536) * 1536, nc = size_t(M) * 1536;
    std::vector<float> h(na); unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    float *A, *W, *C; hipMalloc(&A, na * 4); hipMalloc(&W, nw * 4); hipMalloc(&C, nc * 4);
    hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), nw * 4, hipMemcpyHostToDevice);
    struct { const char* name; int N, K; } shapes[] = {{"qkv  N1152 K384 ", 1152, 384}, {"proj N384  K384 ", 384, 384},
                                                       {"fc1  N1536 K384 ", 1536, 384}, {"fc2  N384  K1536", 384, 1536}};
    for (int round = 0; round < 3; ++round) {
        {   // pure MFMA rate: same grid geometry, 768 MFMAs per wave x rounds
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            const int iters = 192 * 8, blocks = 768 * 4;
            hipEventRecord(a); hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(256), 0, 0, C, iters); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            printf("round %d  mfma-only: %.3f ms %.1f TF\n", round, ms, double(blocks) * 4 * iters * 4 * 4096 / ms / 1e9);
        }
        for (auto& sh : shapes) {
            P g{A, W, C, M, sh.N, sh.K};
            const double gf = 2.0 * M * sh.N * sh.K / 1e9;
            float t0 = run<0, 1>(g, 0), t1 = run<8, 1>(g, 0), t2 = run<0, 0>(g, 0), t3 = run<1, 1>(g, 0),
                  t4 = run<2, 1>(g, 0), t5 = run<3, 0>(g, 0), t6 = run_persistent(g, 3);
            printf("  %s PERSISTENT %.3f (%.1f TF)\n", sh.name, t6, gf / t6);
            printf("  %s buffer-ld %.3f (%.1f TF) | pointer-ld %.3f (%.1f) | noepi %.3f (%.1f) | noload %.3f (%.1f) | nolds %.3f (%.1f) | mfma+ldsread only %.3f (%.1f)\n",
                   sh.name, t0, gf / t0, t1, gf / t1, t2, gf / t2, t3, gf / t3, t4, gf / t4, t5, gf / t5);
        }
    }
    return 0;
}
