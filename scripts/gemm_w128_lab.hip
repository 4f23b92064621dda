// Dev lab (not shipped): the f16x3 planes GEMM mainloop (256 x 256 tile, LDS-direct staging, v_mfma_f32_16x16x32_f16, FC1's shape)
// with TWO register blockings and no epilogue (the accumulators are reduced to a checksum):
//   WAVES = 8: 2 x 4 waves of 128 x 64 (the shipped kernels: two waves per SIMD, 24 fragment reads per 96 MFMAs and K-step)
//   WAVES = 4: 2 x 2 waves of 128 x 128 (one wave per SIMD, 256 accumulator registers, 32 fragment reads per 192 MFMAs: 0.67 x the LDS bytes)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/gemm_w128_lab.hip -o scripts/_lab/gemm_w128_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
#define FENCE() __builtin_amdgcn_sched_barrier(0x76)

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mainloop(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ out,
                                                       int M, int N, int K) {
    constexpr int NWN = WAVES / 2, CB = 256 / NWN / 16;   // column blocks of 16 per wave: 4 or 8
    constexpr int STAGE = 512 * 64;                       // halves per stage: 256 A rows + 256 W rows of 128 B
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN, l15 = lane & 15, q4 = lane >> 4;
    const int tiles_n = N / 256, m0 = (blockIdx.x / tiles_n) * 256, n0 = (blockIdx.x % tiles_n) * 256;
    const unsigned pitch = unsigned(K) * 4u;   // bytes per planes row
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, unsigned(M) * pitch, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(W), 0, unsigned(N) * pitch, 0x00020000);
    const int nk = K / 32, r8 = lane >> 3, piece = (lane & 7) ^ r8;
    constexpr int IPW = 32 / WAVES;   // instructions per wave, operand and K-step (8 rows each)
    auto stage = [&](int s, int k) {
        _Float16* S = lds + s * STAGE;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int row = (wave * IPW + i) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(S + row * 64), 16, unsigned(m0 + row + r8) * pitch + unsigned(piece) * 16u, unsigned(k) * 128u, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(S + (256 + row) * 64), 16, unsigned(n0 + row + r8) * pitch + unsigned(piece) * 16u, unsigned(k) * 128u, 0, 0);
        }
    };
    const int a_row = (wm * 128 + l15) * 64, w_row = (256 + wn * (16 * CB) + l15) * 64;
    const int swz0 = 8 * (q4 ^ (l15 & 7)), swz1 = 8 * ((4 + q4) ^ (l15 & 7));
    f32x4 acc[8][CB];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const _Float16* S = lds + (kt & 1) * STAGE;
        f16x8 wh[CB], wl[CB], ah[3], al[3];
        auto rd = [&](int r) {
            ah[r % 3] = *reinterpret_cast<const f16x8*>(S + a_row + r * (16 * 64) + swz0);
            al[r % 3] = *reinterpret_cast<const f16x8*>(S + a_row + r * (16 * 64) + swz1);
        };
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            wl[c] = *reinterpret_cast<const f16x8*>(S + w_row + c * (16 * 64) + swz1);
            wh[c] = *reinterpret_cast<const f16x8*>(S + w_row + c * (16 * 64) + swz0);
        }
        rd(0);
        rd(1);
        FENCE();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c], ah[r % 3], acc[r][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], al[r % 3], acc[r][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], ah[r % 3], acc[r][c], 0, 0, 0);
            FENCE();
            if (r + 2 < 8) rd(r + 2);
            FENCE();
        }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) s += acc[r][c];
    const float t = (s[0] + s[1]) + (s[2] + s[3]);
    out[(size_t)blockIdx.x * 64 * WAVES + tid] = t;   // one float per thread: 2 KB per tile
}

__global__ void fill(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = _Float16(float(int(x & 2047) - 1024) * (1.f / 512.f));
    }
}

template <int WAVES>
double run(const _Float16* A, const _Float16* W, float* out, int M, int N, int K, double* checksum) {
    const int tiles = (M / 256) * (N / 256);
    hipFuncSetAttribute((const void*)mainloop<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&] { hipLaunchKernelGGL(mainloop<WAVES>, dim3(tiles), dim3(64 * WAVES), 131072, 0, A, W, out, M, N, K); };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    static float host[2048];
    hipMemcpy(host, out, sizeof(float) * 64 * WAVES, hipMemcpyDeviceToHost);
    double cs = 0; for (int i = 0; i < 64 * WAVES; ++i) cs += host[i];
    *checksum = cs;
    return ms;
}

int main() {
    const int M = 97792, N = 1536, K = 384;   // FC1 of the 64-image chunk (382 row tiles: M rounded down to whole tiles)
    _Float16 *A, *W; float* out;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&out, (size_t)(M / 256) * (N / 256) * 512 * 4);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, (size_t)M * K * 2, 1u);
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, W, (size_t)N * K * 2, 2u);
    const double gf = 2.0 * M * N * K * 1e-9;
    for (int rep = 0; rep < 2; ++rep) {
        double c8, c4;
        const double t8 = run<8>(A, W, out, M, N, K, &c8), t4 = run<4>(A, W, out, M, N, K, &c4);
        printf("8 waves of 128 x 64 : %.4f ms  %6.1f TFLOP/s algorithmic (x3 executed: %.0f)  checksum %.6e\n", t8, gf / t8, 3 * gf / t8, c8);
        printf("4 waves of 128 x 128: %.4f ms  %6.1f TFLOP/s algorithmic (x3 executed: %.0f)  checksum %.6e\n", t4, gf / t4, 3 * gf / t4, c4);
    }
    return 0;
}
