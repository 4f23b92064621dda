// "f16x3" NT GEMM: fp32 operands split on the fly into hi + lo f16 halves (x = hi + lo with
// hi = f16(x), lo = f16(x - hi): 22 significand bits), three v_mfma_f32_32x32x16_f16 per product
// block (hi.hi + hi.lo + lo.hi, fp32 accumulate).  This moves the contraction from the f32 MFMA —
// which on gfx950 runs on the VALU lanes at 64 FLOP/clk/SIMD — to the real matrix cores
// (1024 FLOP/clk/SIMD for f16): 16/3 = 5.3x the fp32-MFMA rate at ~2^-21 relative accuracy, and
// the VALU work (splitting, epilogues) overlaps with the matrix pipe instead of stealing from it.
//
// Same interface, tiling (128x128x32, 4 waves x 2x2 32x32 tiles), buffer-load register staging,
// LDS-transposed epilogue and epilogue modes as gemm_f32.hip; fp32 in, fp32 out.
// Both operands are multiplied by a power of two before the split (activations 2^3, weights 2^8 by
// default; exact, undone by one exact multiply in the epilogue) so that the lo half of every
// element that matters stays in the f16 normal range: the representation is relative (2^-22) for
// |a| >= 2^-6 and |w| >= 2^-11 and absolute (2^-28 resp. 2^-33) below — far under the fp32 chain's
// own rounding for O(1) activations.  Range contract: |a| < 8188, |w| < 255 (f16 max / scale).
#include "gemm_core.h"
#include "kernels.h"

namespace {

using namespace gemm_core;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int PL_ST = 40;                       // plane row stride in halves: 80 B = 5 x 16 B (odd) -> b128 reads conflict-free
constexpr int PLANE = 128 * PL_ST;              // halves per plane (128 rows)
constexpr int STAGE = 4 * PLANE;                // A hi, A lo, W hi, W lo
// TWO LDS stages of 40 KB: exactly two workgroups per CU (2 x 80 KB = the CU's 160 KB).  The
// split + ds_write of K-step t+1 goes to the other stage while the MFMAs of K-step t run, so a
// K-step has ONE barrier and its VALU / LDS-write / global-load work sits between its MFMAs (the f16
// MFMA has its own pipe; co-resident waves run in lockstep and would not hide it for each other).
constexpr size_t X3_LDS_BYTES = size_t(2) * STAGE * sizeof(_Float16);
static_assert(X3_LDS_BYTES >= size_t(4) * 32 * EPI_ST * sizeof(float), "epilogue staging must fit");

constexpr float A_SCALE = 8.0f, W_SCALE = 256.0f;  // powers of two: exact

__device__ __forceinline__ void split(f32x4 v, float scale, f16x4& hi, f16x4& lo) {
    pope_split4(v * scale, hi, lo);   // common.h: v_cvt_pk_f16_f32 x2 + v_fma_mixlo/mixhi_f16 x2
}

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf_scalar(float x);  // defined below (shared formula with gemm_f32.hip)
// the same formula on a pair (v_pk_fma_f32 / v_pk_mul_f32 for the polynomial): gemm_f32.hip:gelu_erf2
__device__ __forceinline__ f32x2 gelu_erf_pair(f32x2 x) {
    constexpr float P = 0.3275911f * 0.70710678118654752440f;
    constexpr float A1 = 0.5f * 0.254829592f, A2 = 0.5f * -0.284496736f, A3 = 0.5f * 1.421413741f,
                    A4 = 0.5f * -1.453152027f, A5 = 0.5f * 1.061405429f;
    constexpr float NHL2E = -0.5f * 1.44269504088896340736f;
    f32x2 t, e, relu;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        t[i] = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x[i]), P, 1.0f));
        relu[i] = __builtin_fmaxf(x[i], 0.0f);
    }
    const f32x2 arg = (x * NHL2E) * x;
    e[0] = __builtin_amdgcn_exp2f(arg[0]);
    e[1] = __builtin_amdgcn_exp2f(arg[1]);
    f32x2 poly = __builtin_elementwise_fma(t, f32x2{A5, A5}, f32x2{A4, A4});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A3, A3});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A2, A2});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A1, A1});
    const f32x2 q = (poly * t) * e;
    return __builtin_elementwise_fma(relu, __builtin_elementwise_fma(q, f32x2{-2.f, -2.f}, f32x2{1.f, 1.f}), x * q);
}

template <int EPI>
__global__ __launch_bounds__(THREADS, 2) void gemm_nt_f16x3_kernel(const GemmParams g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;

    const BufferLoader la(g.A, g.M, g.lda, m0), lw(g.W, g.N, g.ldw, n0);
    const int nk = g.K / BK;
    // One register staging set.  During K-step t it holds K-step t+1 (split into the other LDS stage)
    // and is then refilled with K-step t+2, which has a whole K-step plus a barrier to arrive.
    f32x4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = la.load(i, 0);
        rw[i] = lw.load(i, 0);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    auto stage_write = [&](int st) {
        _Float16* Ah = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 hi, lo;
            const int o = (srow + 32 * i) * PL_ST + scol;
            split(ra[i], A_SCALE, hi, lo);
            *reinterpret_cast<f16x4*>(Ah + o) = hi;
            *reinterpret_cast<f16x4*>(Ah + PLANE + o) = lo;
            split(rw[i], W_SCALE, hi, lo);
            *reinterpret_cast<f16x4*>(Ah + 2 * PLANE + o) = hi;
            *reinterpret_cast<f16x4*>(Ah + 3 * PLANE + o) = lo;
        }
    };
    const int a_off = (wm * 64 + r) * PL_ST + 8 * h, w_off = 2 * PLANE + (wn * 64 + r) * PL_ST + 8 * h;
    // K-step kt: MFMAs on stage kt&1; meanwhile the staged registers (K-step kt+1) are split into the
    // other stage and then refilled with K-step kt+2.
    // MODE 2: steady state (split + refill), 1: split only (K-step kt+2 does not exist), 0: last K-step.
    // The variants are separate straight-line bodies so that each K-step is ONE basic block and the
    // scheduler can interleave its MFMAs with the split / LDS / load instructions.
    auto kstep = [&](int kt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const _Float16* S = lds + (kt & 1) * STAGE;
        if constexpr (MODE >= 1) stage_write((kt + 1) & 1);
        if constexpr (MODE == 2) {
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
                ah[t] = *reinterpret_cast<const f16x8*>(S + a_off + t * 32 * PL_ST + kg * 16);
                al[t] = *reinterpret_cast<const f16x8*>(S + PLANE + a_off + t * 32 * PL_ST + kg * 16);
                wh[t] = *reinterpret_cast<const f16x8*>(S + w_off + t * 32 * PL_ST + kg * 16);
                wl[t] = *reinterpret_cast<const f16x8*>(S + PLANE + w_off + t * 32 * PL_ST + kg * 16);
            }
            // accumulators hold C^T (rows over n): A-operand = W fragment, B-operand = A fragment
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = mfma_f16(wl[ni], ah[mi], acc[mi][ni]);  // small terms first
                    acc[mi][ni] = mfma_f16(wh[ni], al[mi], acc[mi][ni]);
                    acc[mi][ni] = mfma_f16(wh[ni], ah[mi], acc[mi][ni]);
                }
        }
        // 24 MFMAs : ~130 VALU (split) : 16 LDS reads : 16 LDS writes : 8 buffer loads — pin an even mix
        // (LLVM sched groups: 0x8 MFMA, 0x2 VALU, 0x100 DS read, 0x200 DS write, 0x20 VMEM read)
        if constexpr (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                if (i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i >= 4 && i < 20) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (MODE == 2 && i >= 12 && i < 20) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __syncthreads();  // stage (kt+1)&1 is published, stage kt&1 is free
    };

    stage_write(0);  // K-step 0
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = la.load(i, BK);
            rw[i] = lw.load(i, BK);
        }
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) kstep(kt, std::integral_constant<int, 2>{});
        else if (kt + 1 < nk) kstep(kt, std::integral_constant<int, 1>{});
        else kstep(kt, std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = acc[mi][ni] * (1.0f / (A_SCALE * W_SCALE));

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
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_nt_f16x3_kernel<EPI>, X3_LDS_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nt_f16x3_kernel<EPI>), dim3(tiles), dim3(THREADS), X3_LDS_BYTES, stream, g);
    return pope_check_launch();
}

}  // namespace

// Batched similarity for the dense matcher: C[b] = (A[b] . W[b]^T * alpha) / divisor on planes operands.
int pope_launch_sim_f16x3_planes(const GemmParams& g, hipStream_t stream) {
    if (g.epilogue != EPI_SIM || !g.a_pl || !g.w_pl || !g.C || g.nbatch <= 0 || g.M <= 0 || g.N <= 0) return POPE_ERR_ARG;
    if (g.K < 2 * BK || (g.K % BK) || (g.lda & 31) || (g.ldw & 31) || g.ldc != g.N || g.divisor_eff == 0.f) return POPE_ERR_ARG;
    if ((g.row_part || g.col_pmax) &&
        (!g.row_part || !g.col_pmax || !g.col_psum || g.ncb != 2 * ((g.N + BN - 1) / BN) || g.nrb != 4 * ((g.M + BM - 1) / BM) ||
         g.ldp < g.N || (g.ldp & 3)))
        return POPE_ERR_ARG;
    if ((size_t(g.nbatch) * g.M + BM) * g.lda * 4 >= (size_t(1) << 32) || (size_t(g.nbatch) * g.N + BN) * g.ldw * 4 >= (size_t(1) << 32) ||
        (size_t(g.nbatch) * g.M + BM) * g.ldc * 4 >= (size_t(1) << 32) - 512)
        return POPE_ERR_ARG;
    if (static_cast<long long>(g.nbatch) * ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) > 0x7fffffffLL) return POPE_ERR_ARG;
    return pope_launch_planes16(g, stream);
}

int pope_launch_gemm_nt_f16x3_planes(const GemmParams& g, hipStream_t stream) {
    static_assert(A_SCALE == K_PLANES_ACT_SCALE && W_SCALE == K_PLANES_W_SCALE, "plane scales");
    if (g.M <= 0 || g.N <= 0 || g.K < 2 * BK || (g.K % BK) || (g.N & 3) || (g.ldc & 3) || (g.lda & 7) || (g.ldw & 7)) return POPE_ERR_ARG;
    if (!g.a_pl || !g.w_pl || (g.lda & 31) || (g.ldw & 31)) return POPE_ERR_ARG;
    if (size_t(g.M + 256) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N + BN) * g.ldw * 4 >= (size_t(1) << 32)) return POPE_ERR_ARG;
    const bool out_planes = g.c_pl != nullptr;
    if (out_planes ? (g.ldc & 31) != 0 : !g.C) return POPE_ERR_ARG;
    // the epilogue addresses C (and res) through 32-bit buffer offsets
    if (size_t(g.M + 256) * g.ldc * 4 >= (size_t(1) << 32) - 512) return POPE_ERR_ARG;
    if (g.epilogue == EPI_BIAS_LS_RES && size_t(g.M + 256) * g.ldres * 4 >= (size_t(1) << 32) - 512) return POPE_ERR_ARG;
    switch (g.epilogue) {
        case EPI_BIAS:
        case EPI_BIAS_GELU:
        case EPI_BIAS_RELU: return pope_launch_planes16(g, stream);
        case EPI_QKV_F16:
            if (!g.plain || !out_planes || g.sam_dim <= 0 || (g.sam_dim & 63)) return POPE_ERR_ARG;
            return pope_launch_planes16(g, stream);
        case EPI_BIAS_LS_RES:
            if (!g.res || out_planes || (!g.gamma && g.res_mod <= 0)) return POPE_ERR_ARG;
            return pope_launch_planes16(g, stream);
    }
    return POPE_ERR_ARG;
}

bool pope_gemm_f16x3_supported(const GemmParams& g) {
    return g.epilogue != EPI_POSB && (g.K % BK) == 0 && size_t(g.M + BM) * g.lda * 4 < (size_t(1) << 32) &&
           size_t(g.N + BN) * g.ldw * 4 < (size_t(1) << 32);
}

int pope_launch_gemm_nt_f16x3(const GemmParams& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.N & 3) || (g.ldc & 3) || !pope_gemm_f16x3_supported(g)) return POPE_ERR_ARG;
    if (g.range_flag) {  // this kernel splits fp32 operands inside its K loop: check them in a scan of their own
        if (g.lda != g.K || g.ldw != g.K) return POPE_ERR_ARG;
        int rc = pope_launch_range_check(g.A, size_t(g.M) * g.K, A_SCALE, g.range_flag, POPE_RANGE_INPUT, stream);
        if (!rc) rc = pope_launch_range_check(g.W, size_t(g.N) * g.K, W_SCALE, g.range_flag, POPE_RANGE_INPUT, stream);
        if (rc) return rc;
    }
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
