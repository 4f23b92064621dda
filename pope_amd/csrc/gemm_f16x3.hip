// "f16x3" NT GEMM: fp32 operands split on the fly into hi + lo f16 halves (x = hi + lo with
// hi = f16(x), lo = f16(x - hi): 22 significand bits), three v_mfma_f32_32x32x16_f16 per product
// block (hi.hi + hi.lo + lo.hi, fp32 accumulate).  This moves the contraction from the f32 MFMA —
// which on gfx950 runs on the VALU lanes at 64 FLOP/clk/SIMD — to the real matrix cores
// (1024 FLOP/clk/SIMD for f16): 16/3 = 5.3x the fp32-MFMA rate at ~2^-21 relative accuracy, and
// the VALU work (splitting, epilogues) overlaps with the matrix pipe instead of stealing from it.
//
// Same interface, tiling (128x128x32, 4 waves x 2x2 32x32 tiles), buffer-load register staging,
// LDS-transposed epilogue and epilogue modes as gemm_f32.hip; fp32 in, fp32 out.
// The lo halves are stored scaled by 2^11 (lo' = f16((x - hi) * 2048), same magnitude as hi, so they
// never fall into the f16 subnormal range) and the two cross terms hi.lo' + lo'.hi accumulate in
// their own fp32 accumulator, folded in as acc + 2^-11 * cross in the epilogue: the representation
// stays relative (2^-22) for every |x| in [6.1e-5, 65504).  Range contract: |operand| < 65504.
#include "gemm_core.h"
#include "kernels.h"

namespace {

using namespace gemm_core;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int PL_ST = 40;                       // plane row stride in halves: 80 B = 5 x 16 B (odd) -> b128 reads conflict-free
constexpr int PLANE = 128 * PL_ST;              // halves per plane (128 rows)
constexpr size_t X3_LDS_BYTES = size_t(4) * PLANE * sizeof(_Float16);  // A hi, A lo, W hi, W lo = 40 KB
static_assert(X3_LDS_BYTES >= size_t(4) * 32 * EPI_ST * sizeof(float), "epilogue staging must fit");

__device__ __forceinline__ void split(f32x4 v, f16x4& hi, f16x4& lo) {
    hi = __builtin_convertvector(v, f16x4);                          // v_cvt_pk_f16_f32 x2 (RNE)
#ifdef X3_FAKE_SPLIT  // timing experiment only: drop the lo computation (wrong results)
    lo = hi;
#else
    lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * 2048.0f, f16x4);
#endif
}

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf_scalar(float x);  // defined below (shared formula with gemm_f32.hip)

template <int EPI>
__global__ __launch_bounds__(THREADS, 2) void gemm_nt_f16x3_kernel(const GemmParams g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Ah = reinterpret_cast<_Float16*>(smem);
    _Float16* Al = Ah + PLANE;
    _Float16* Wh = Al + PLANE;
    _Float16* Wl = Wh + PLANE;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;

    const BufferLoader la(g.A, g.M, g.lda, m0), lw(g.W, g.N, g.ldw, n0);
    // Two register staging sets, prefetch distance TWO K-steps: a K-step's 24 MFMAs last only
    // ~770 cycles, far less than an L2/HBM round trip, so the loads of K-step t+2 are issued as soon
    // as the registers of K-step t have been split into LDS (out-of-range K-steps read zeros).
    f32x4 ra0[4], rw0[4], ra1[4], rw1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra0[i] = la.load(i, 0);
        rw0[i] = lw.load(i, 0);
    }
    const int nk = g.K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra1[i] = la.load(i, nk > 1 ? BK : 0);
        rw1[i] = lw.load(i, nk > 1 ? BK : 0);
    }
    f32x16 acc[2][2], cross[2][2];  // hi.hi | hi.lo' + lo'.hi (scaled by 2^11)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[mi][ni][i] = 0.f; cross[mi][ni][i] = 0.f; }

    const int a_off = (wm * 64 + r) * PL_ST + 8 * h, w_off = (wn * 64 + r) * PL_ST + 8 * h;
    auto kstep = [&](int kt, f32x4 (&ra)[4], f32x4 (&rw)[4]) {
        __syncthreads();  // every wave is done reading the previous stage
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 hi, lo;
            const int o = (srow + 32 * i) * PL_ST + scol;
            split(ra[i], hi, lo);
            *reinterpret_cast<f16x4*>(Ah + o) = hi;
            *reinterpret_cast<f16x4*>(Al + o) = lo;
            split(rw[i], hi, lo);
            *reinterpret_cast<f16x4*>(Wh + o) = hi;
            *reinterpret_cast<f16x4*>(Wl + o) = lo;
        }
        __syncthreads();
        if (kt + 2 < nk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = la.load(i, (kt + 2) * BK);
                rw[i] = lw.load(i, (kt + 2) * BK);
            }
        }
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            f16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(Ah + a_off + t * 32 * PL_ST + kg * 16);
                al[t] = *reinterpret_cast<const f16x8*>(Al + a_off + t * 32 * PL_ST + kg * 16);
                wh[t] = *reinterpret_cast<const f16x8*>(Wh + w_off + t * 32 * PL_ST + kg * 16);
                wl[t] = *reinterpret_cast<const f16x8*>(Wl + w_off + t * 32 * PL_ST + kg * 16);
            }
            // accumulators hold C^T (rows over n): A-operand = W fragment, B-operand = A fragment
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    cross[mi][ni] = mfma_f16(wl[ni], ah[mi], cross[mi][ni]);
                    cross[mi][ni] = mfma_f16(wh[ni], al[mi], cross[mi][ni]);
                    acc[mi][ni] = mfma_f16(wh[ni], ah[mi], acc[mi][ni]);
                }
        }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, ra0, rw0);
        if (kt + 1 < nk) kstep(kt + 1, ra1, rw1);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = acc[mi][ni] + cross[mi][ni] * (1.0f / 2048.0f);

    epilogue_rows(acc, smem, [&](int tr, int tc, f32x4 v) {
        const int row = m0 + tr, col = n0 + tc;
        if (row >= g.M || col >= g.N) return;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias = *reinterpret_cast<const f32x4*>(g.bias + col);
        if constexpr (EPI == EPI_BIAS) {
            v = v + bias;
        } else if constexpr (EPI == EPI_BIAS_GELU) {
            v = v + bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf_scalar(v[e]);
        } else {
            const f32x4 gamma = *reinterpret_cast<const f32x4*>(g.gamma + col);
            const f32x4 res = *reinterpret_cast<const f32x4*>(g.res + size_t(row) * g.ldres + col);
            v = res + (v + bias) * gamma;
        }
        *reinterpret_cast<f32x4*>(g.C + size_t(row) * g.ldc + col) = v;
    });
}

// exact-erf GELU, Abramowitz-Stegun 7.1.26 form (see gemm_f32.hip:gelu_erf2 for the derivation)
__device__ __forceinline__ float gelu_erf_scalar(float x) {
    constexpr float P = 0.3275911f * 0.70710678118654752440f;
    constexpr float A1 = 0.5f * 0.254829592f, A2 = 0.5f * -0.284496736f, A3 = 0.5f * 1.421413741f,
                    A4 = 0.5f * -1.453152027f, A5 = 0.5f * 1.061405429f;
    constexpr float NHL2E = -0.5f * 1.44269504088896340736f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x), P, 1.0f));
    const float e = __builtin_amdgcn_exp2f((x * NHL2E) * x);
    float poly = __builtin_fmaf(t, A5, A4);
    poly = __builtin_fmaf(poly, t, A3);
    poly = __builtin_fmaf(poly, t, A2);
    poly = __builtin_fmaf(poly, t, A1);
    const float q = (poly * t) * e;
    return __builtin_fmaf(__builtin_fmaxf(x, 0.0f), __builtin_fmaf(q, -2.f, 1.f), x * q);
}

template <int EPI>
int launch(const GemmParams& g, hipStream_t stream) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nt_f16x3_kernel<EPI>), dim3(tiles), dim3(THREADS), X3_LDS_BYTES, stream, g);
    return pope_check_launch();
}

}  // namespace

bool pope_gemm_f16x3_supported(const GemmParams& g) {
    return g.epilogue != EPI_POSB && (g.K % BK) == 0 && size_t(g.M + BM) * g.lda * 4 < (size_t(1) << 32) &&
           size_t(g.N + BN) * g.ldw * 4 < (size_t(1) << 32);
}

int pope_launch_gemm_nt_f16x3(const GemmParams& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.N & 3) || (g.ldc & 3) || !pope_gemm_f16x3_supported(g)) return POPE_ERR_ARG;
    if ((g.lda & 3) || (g.ldw & 3) || (reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.W) & 15) ||
        (reinterpret_cast<uintptr_t>(g.C) & 15))
        return POPE_ERR_ARG;
    switch (g.epilogue) {
        case EPI_BIAS: return launch<EPI_BIAS>(g, stream);
        case EPI_BIAS_GELU: return launch<EPI_BIAS_GELU>(g, stream);
        case EPI_BIAS_LS_RES:
            if (!g.gamma || !g.res) return POPE_ERR_ARG;
            return launch<EPI_BIAS_LS_RES>(g, stream);
    }
    return POPE_ERR_ARG;
}
