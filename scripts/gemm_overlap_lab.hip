// Dev lab (not shipped): can a wave work a finished tile's epilogue off BETWEEN the MFMAs of its next tile?  The planes GEMM
// mainloop of FC1's shape as a persistent workgroup of 4 waves (one per SIMD; tile 256 x 128, a wave owns 128 x 64 = 128
// accumulator registers) with TWO accumulator sets:
//   MODE 0: K loops only (the accumulators are reduced to a checksum)
//   MODE 1: K loop, then the tile's epilogue (exact-erf GELU on every value + f16 stores straight from the MFMA layout) — today's order
//   MODE 2: the epilogue of tile i is cut into 96 slices and issued between the MFMA groups of tile i + 1
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/gemm_overlap_lab.hip -o scripts/_lab/gemm_overlap_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
#define FENCE() __builtin_amdgcn_sched_barrier(0x76)

__device__ __forceinline__ f32x2 gelu_pair(f32x2 x) {   // gemm_plain.hip: pl_gelu_pair
    constexpr float P = 0.3275911f * 0.70710678118654752440f;
    constexpr float A1 = 0.5f * 0.254829592f, A2 = 0.5f * -0.284496736f, A3 = 0.5f * 1.421413741f, A4 = 0.5f * -1.453152027f, A5 = 0.5f * 1.061405429f;
    constexpr float NHL2E = -0.5f * 1.44269504088896340736f;
    f32x2 t, e, relu;
    for (int i = 0; i < 2; ++i) { t[i] = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x[i]), P, 1.0f)); relu[i] = __builtin_fmaxf(x[i], 0.0f); }
    const f32x2 arg = (x * NHL2E) * x;
    e[0] = __builtin_amdgcn_exp2f(arg[0]); e[1] = __builtin_amdgcn_exp2f(arg[1]);
    f32x2 poly = __builtin_elementwise_fma(t, f32x2{A5, A5}, f32x2{A4, A4});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A3, A3});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A2, A2});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A1, A1});
    const f32x2 q = (poly * t) * e;
    return __builtin_elementwise_fma(relu, __builtin_elementwise_fma(q, f32x2{-2.f, -2.f}, f32x2{1.f, 1.f}), x * q);
}

#define GELU_PAIR(v) gelu_pair(v)
#ifndef PER_GAP
#define PER_GAP 2
#endif
constexpr int NK = 12;   // K = 384

template <int MODE>
__global__ __launch_bounds__(256) void overlap(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ out,
                                               float* __restrict__ sums, int M, int N, int n_tiles) {
    constexpr int STAGE = 384 * 64;   // halves: 256 A rows + 128 W rows of 128 B
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, q4 = lane >> 4;
    const int tiles_n = N / 128;
    const unsigned pitch = unsigned(NK * 32) * 4u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, unsigned(M) * pitch, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(W), 0, unsigned(N) * pitch, 0x00020000);
    const int r8 = lane >> 3, piece = (lane & 7) ^ r8;
    auto stage = [&](int s, int tile, int k) {
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 128;
        _Float16* S = lds + s * STAGE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (wave * 8 + i) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(S + row * 64), 16, unsigned(m0 + row + r8) * pitch + unsigned(piece) * 16u, unsigned(k) * 128u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(S + (256 + row) * 64), 16, unsigned(n0 + row + r8) * pitch + unsigned(piece) * 16u, unsigned(k) * 128u, 0, 0);
        }
    };
    const int a_row = (wm * 128 + l15) * 64, w_row = (256 + wn * 64 + l15) * 64;
    const int swz0 = 8 * (q4 ^ (l15 & 7)), swz1 = 8 * ((4 + q4) ^ (l15 & 7));
    f32x4 acc0[8][4], acc1[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc0[r][c] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[r][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float checksum = 0.f;
    // epilogue of one accumulator register (4 values of one row block / column block): GELU + f16 store from the MFMA layout
    auto epi_reg = [&](f32x4 v, int tile, int idx) {
        const f32x2 g0 = GELU_PAIR((f32x2{v[0], v[1]})), g1 = GELU_PAIR((f32x2{v[2], v[3]}));
        const f16x4 h = __builtin_convertvector(f32x4{g0[0], g0[1], g1[0], g1[1]} * 8.0f, f16x4);
        *reinterpret_cast<f16x4*>(out + (((size_t)tile * 256 + tid) * 32 + idx) * 4) = h;
    };
    int stage_i = 0;
    auto tile_pass = [&](f32x4 (&cur)[8][4], f32x4 (&prev)[8][4], int tile, int prev_tile, int next_tile) {
        // prologue of this tile's K loop was issued by the previous pass (or before the loop)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) cur[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (kt + 1 < NK) stage(stage_i ^ 1, tile, kt + 1);
            else if (next_tile < n_tiles) stage(stage_i ^ 1, next_tile, 0);
            const _Float16* S = lds + stage_i * STAGE;
            stage_i ^= 1;
            f16x8 wh[4], wl[4], ah[3], al[3];
            auto rd = [&](int r) {
                ah[r % 3] = *reinterpret_cast<const f16x8*>(S + a_row + r * (16 * 64) + swz0);
                al[r % 3] = *reinterpret_cast<const f16x8*>(S + a_row + r * (16 * 64) + swz1);
            };
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                wl[c] = *reinterpret_cast<const f16x8*>(S + w_row + c * (16 * 64) + swz1);
                wh[c] = *reinterpret_cast<const f16x8*>(S + w_row + c * (16 * 64) + swz0);
            }
            rd(0);
            rd(1);
            FENCE();
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int c = 0; c < 4; ++c) cur[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c], ah[r % 3], cur[r][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) cur[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], al[r % 3], cur[r][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) cur[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], ah[r % 3], cur[r][c], 0, 0, 0);
                if constexpr (MODE == 2) {   // slice u = kt * 8 + r of the previous tile's epilogue: one GELU pair per unit for 64 units,
                    const int u = kt * 8 + r;   // the register's conversion + store behind its second pair
                    if (u < 64) {
                        f32x4& v = prev[(u >> 1) >> 2][(u >> 1) & 3];
                        const f32x2 gin = (u & 1) ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]};
                        const f32x2 gp = GELU_PAIR(gin);
                        if (u & 1) {
                            const f16x4 h4 = __builtin_convertvector(f32x4{v[0], v[1], gp[0], gp[1]} * 8.0f, f16x4);
                            *reinterpret_cast<f16x4*>(out + (((size_t)(prev_tile < 0 ? tile : prev_tile) * 256 + tid) * 32 + (u >> 1)) * 4) = h4;
                        } else {
                            v[0] = gp[0]; v[1] = gp[1];
                        }
                        // one MFMA, then two vector instructions, twelve times: the slice rides between the MFMAs of the unit
#pragma unroll
                        for (int i = 0; i < 12; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, PER_GAP, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // the slice stays in its unit
                }
                FENCE();
                if (r + 2 < 8) rd(r + 2);
                FENCE();
            }
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 32; ++u) epi_reg(cur[u >> 2][u & 3], tile, u);
        }
        if constexpr (MODE == 0) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 32; ++u) s += cur[u >> 2][u & 3];
            checksum += (s[0] + s[1]) + (s[2] + s[3]);
        }
    };
    int tile = blockIdx.x, prev_tile = -1;
    if (tile < n_tiles) stage(0, tile, 0);
    while (tile < n_tiles) {
        int nxt = tile + gridDim.x;
        tile_pass(acc0, acc1, tile, prev_tile, nxt);
        prev_tile = tile; tile = nxt;
        if (tile >= n_tiles) { if constexpr (MODE == 2) { for (int u = 0; u < 32; ++u) epi_reg(acc0[u >> 2][u & 3], prev_tile, u); } break; }
        nxt = tile + gridDim.x;
        tile_pass(acc1, acc0, tile, prev_tile, nxt);
        prev_tile = tile; tile = nxt;
        if (tile >= n_tiles) { if constexpr (MODE == 2) { for (int u = 0; u < 32; ++u) epi_reg(acc1[u >> 2][u & 3], prev_tile, u); } break; }
    }
    if (MODE == 0) sums[blockIdx.x * 256 + tid] = checksum;
}

__global__ void fill(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = _Float16(float(int(x & 2047) - 1024) * (1.f / 512.f));
    }
}

template <int MODE>
float run(const _Float16* A, const _Float16* W, _Float16* out, float* sums, int M, int N, int tiles) {
    hipFuncSetAttribute((const void*)overlap<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&] { hipLaunchKernelGGL(overlap<MODE>, dim3(256), dim3(256), 98304, 0, A, W, out, sums, M, N, tiles); };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}

int main() {
    const int M = 97792, N = 1536, K = 384, tiles = (M / 256) * (N / 128);
    _Float16 *A, *W, *out; float* sums;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&out, (size_t)tiles * 256 * 32 * 8); hipMalloc(&sums, 256 * 256 * 4);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, (size_t)M * K * 2, 1u);
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, W, (size_t)N * K * 2, 2u);
    const double gf = 2.0 * M * N * K * 1e-9;
    for (int rep = 0; rep < 2; ++rep) {
        const float t0 = run<0>(A, W, out, sums, M, N, tiles), t1 = run<1>(A, W, out, sums, M, N, tiles);
        static _Float16 h1[4096], h2[4096];
        hipMemcpy(h1, out + (size_t)(tiles - 1) * 256 * 32 * 4, sizeof(h1), hipMemcpyDeviceToHost);
        hipMemset(out, 0, (size_t)tiles * 256 * 32 * 8);
        const float t2 = run<2>(A, W, out, sums, M, N, tiles);
        hipMemcpy(h2, out + (size_t)(tiles - 1) * 256 * 32 * 4, sizeof(h2), hipMemcpyDeviceToHost);
        int diff = 0; for (int i = 0; i < 4096; ++i) diff += float(h1[i]) != float(h2[i]);
        printf("K loops only %.4f ms (%.0f TF/s alg) | + epilogue after the loop %.4f ms | epilogue between the next tile's MFMAs %.4f ms | output words differing: %d\n",
               t0, gf / t0, t1, t2, diff);
    }
    return 0;
}
