// fp32 "NT" GEMM on the gfx950 f32 MFMA:  C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogue).
//
// This one kernel family carries every Linear of the DINOv2 block — QKV, proj, FC1, FC2
// (reference: dinov2/dinov2/layers/attention.py:51,60; mlp.py:35-44; layer_scale.py:27-28;
// block.py:105-106) — and, with the im2col gather loader, the patch-embed conv
// (patch_embed.py:69-82) fused with "+cls, +pos_embed" (vision_transformer.py:191-200).
//
// Tiling: 128x128 block tile, BK=32, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles,
// v_mfma_f32_32x32x2_f32 (exact f32 fma chain), three workgroups per CU.  Both operands are
// K-contiguous, so both are staged identically: global -> registers (prefetch of K-step t+1
// in flight under the MFMAs of K-step t) -> one LDS stage, rows padded to 36 floats
// (ds_read_b128 conflict-free: 9*r mod 16 is a bijection on a 16-lane group).  The k order inside an 8-wide group is permuted
// (lane half h takes k = 8j+4h+s at step s) so that one ds_read_b128 feeds four MFMAs; A and
// W use the same permutation, so the sum is over the same products.
#include "gemm_core.h"
#include "kernels.h"

namespace {

using namespace gemm_core;

// nn.GELU() default = exact erf form, 0.5 x (1 + erf(x / sqrt 2)), evaluated
// branch-free on element PAIRS with packed f32 math: the f32 MFMA shares the SIMD's VALU lanes, so
// the epilogue's instruction count is paid in full against the matrix stream (ocml erff: ~35 VALU
// per element with divergent range branches = ~20 % of an FC1 tile).  erfc via Abramowitz-Stegun
// 7.1.26 (|err| <= 1.5e-7): with z = |x|/sqrt2, t = 1/(1 + p z), q = 0.5 t P(t) exp(-z^2):
//   gelu(x) = x (1 - q) for x >= 0,  x q for x < 0   ==   relu(x) (1 - 2q) + x q.
// Max abs error vs the exact form over [-12, 12] in fp32: 4.7e-7 (one ulp at |x| ~ 4).
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    constexpr float P = 0.3275911f * 0.70710678118654752440f;
    constexpr float A1 = 0.5f * 0.254829592f, A2 = 0.5f * -0.284496736f, A3 = 0.5f * 1.421413741f,
                    A4 = 0.5f * -1.453152027f, A5 = 0.5f * 1.061405429f;
    constexpr float NHL2E = -0.5f * 1.44269504088896340736f;  // exp(-x^2/2) = exp2(x * x * NHL2E)
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

enum { LOAD_GENERIC = 0, LOAD_BUFFER = 1, LOAD_PATCH = 2, LOAD_CONV = 3 };

template <int LOADER>
__device__ __forceinline__ f32x4 load_a(const GemmParams& g, int row, int k) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= g.M || k >= g.K) return v;
    if constexpr (LOADER == LOAD_CONV) {
        // implicit 3x3 stride-1 convolution over a zero-bordered NHWC fp32 tensor (conv.hip, the fp32 twin of gemm_planes.hip's
        // CONV loader): K-step (tap, channel c) of output row R reads row R + dy * Wp + dx, channel c
        const int tap = k / g.lda, c = k - tap * g.lda;
        const int dy = tap / 3, dx = tap - 3 * dy;
        return *reinterpret_cast<const f32x4*>(g.A + (size_t(row) + size_t(dy) * g.conv_wp + dx) * g.lda + c);
    } else if constexpr (LOADER != LOAD_PATCH) {
        return *reinterpret_cast<const f32x4*>(g.A + size_t(row) * g.lda + k);
    } else {
        // im2col on the fly: row = (image b, token n); token 0 is the cls slot (zero row, the
        // epilogue adds cls+pos[0]); k = (c, i, j) over the patch, matching the flattened conv
        // weight [dim][3][P][P].
        const int b = row / g.ntok, n = row - b * g.ntok;
        if (n == 0) return v;
        const int p = n - 1, ph = p / g.grid_w, pw = p - ph * g.grid_w;
        const int pp = g.patch * g.patch;
        const float* img = g.A + size_t(b) * 3 * g.img_h * g.img_w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = k + e;
            if (kk < g.K) {
                const int c = kk / pp, rem = kk - c * pp, i = rem / g.patch, j = rem - i * g.patch;
                v[e] = img[(size_t(c) * g.img_h + ph * g.patch + i) * g.img_w + pw * g.patch + j];
            }
        }
        return v;
    }
}

__device__ __forceinline__ f32x4 load_w(const GemmParams& g, int row, int k) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= g.N || k >= g.K) return v;
    return *reinterpret_cast<const f32x4*>(g.W + size_t(row) * g.ldw + k);
}

// Coalesced epilogue through a per-wave 32x68-float LDS transposition (as gemm_core::epilogue_rows), written so that
// no load sits between two stores: bias / gamma are per-lane constants of the tile and fetched once; the row-indexed
// operands (residual, patch-embed table) are fetched eight rows at a time before the eight stores; rows >= M and
// columns >= N are dropped by the buffer descriptor.  (With a load inside each of the 16 row pieces the compiler has to
// wait for vmcnt(0) — every earlier store acknowledged — before each store: DESIGN.md §4 finding 7.)
template <int EPI>
__device__ __forceinline__ void tile_epilogue(const GemmParams& g, int m0, int n0, const f32x16 (&acc)[2][2], float* smem) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4e;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int c4 = (lane & 15) * 4, lr = lane >> 4;
    const int col = n0 + wn * 64 + c4;
    const bool col_ok = col < g.N;
    const int colc = col_ok ? col : 0;
    constexpr unsigned DROP = 0xFFFFFF00u;
    const bool wide = size_t(g.M + BM) * g.ldc * 4 < (size_t(1) << 32) - 512;  // 32-bit buffer offsets cover C
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, gamma = {1.f, 1.f, 1.f, 1.f};
    if constexpr (EPI != EPI_POSB)
        if (g.bias) bias = *reinterpret_cast<const f32x4*>(g.bias + colc);
    if constexpr (EPI == EPI_BIAS_LS_RES) gamma = *reinterpret_cast<const f32x4*>(g.gamma + colc);
    const bool conv_res = EPI == EPI_CONV && g.res != nullptr;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, wide ? unsigned(g.M) * unsigned(g.ldc) * 4u : 0u, 0x00020000);
    __syncthreads();  // all waves have finished reading the last K-step stage
    float* E = smem + wave * 32 * EPI_ST;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g4 + e];
                *reinterpret_cast<f32x4*>(&E[r * EPI_ST + ni * 32 + 8 * g4 + 4 * h]) = v;
            }
        const int row0 = m0 + wm * 64 + mi * 32 + lr;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 extra[4];  // residual / table rows of this half pass (four at a time: the fp32 kernels run at 168 VGPRs)
            if constexpr (EPI == EPI_BIAS_LS_RES || EPI == EPI_POSB || EPI == EPI_CONV) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + 4 * (4 * half + i);
                    extra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (row < g.M && col_ok) {
                        if constexpr (EPI == EPI_BIAS_LS_RES) extra[i] = *reinterpret_cast<const f32x4*>(g.res + size_t(row) * g.ldres + col);
                        else if constexpr (EPI == EPI_CONV) { if (conv_res) extra[i] = *reinterpret_cast<const f32x4*>(g.res + size_t(row) * g.ldres + col); }
                        else extra[i] = *reinterpret_cast<const f32x4*>(g.posb + size_t(row % g.ntok) * g.N + col);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * (4 * half + i);
                f32x4 v = *reinterpret_cast<const f32x4*>(&E[(lr + 4 * (4 * half + i)) * EPI_ST + c4]);
                if constexpr (EPI == EPI_BIAS) {
                    v = v + bias;
                } else if constexpr (EPI == EPI_BIAS_GELU) {
                    v = v + bias;
                    const f32x2 lo = gelu_erf2(f32x2{v[0], v[1]}), hi = gelu_erf2(f32x2{v[2], v[3]});
                    v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else if constexpr (EPI == EPI_BIAS_LS_RES) {
                    v = extra[i] + (v + bias) * gamma;
                } else if constexpr (EPI == EPI_CONV) {   // act(acc + bias [+ shortcut]), act(v) = max(v, 0) + slope * min(v, 0)
                    v = v + bias + extra[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(v[e], 0.f) + g.act_slope * __builtin_fminf(v[e], 0.f);
                } else {  // EPI_POSB: + (pos_embed + conv bias | cls) table indexed by token
                    v = v + extra[i];
                }
                if (wide) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4e, v), rc,
                                                           col_ok ? unsigned(row) * unsigned(g.ldc) * 4u + unsigned(col) * 4u : DROP, 0, 0);
                } else if (row < g.M && col_ok) {
                    *reinterpret_cast<f32x4*>(g.C + size_t(row) * g.ldc + col) = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// One tile per workgroup (hardware dispatcher refills the three slots per CU as tiles retire).
template <int EPI, int LOADER>
__global__ __launch_bounds__(THREADS, LOADER == LOAD_PATCH ? 2 : 3) void gemm_nt_f32_kernel(const GemmParams g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    f32x16 acc[2][2];
    if constexpr (LOADER == LOAD_BUFFER) {
        mainloop(BufferLoader(g.A, g.M, g.lda, m0), BufferLoader(g.W, g.N, g.ldw, n0), g.K, smem, acc);
    } else {
        mainloop(fn_loader([&](int row, int k) { return load_a<LOADER>(g, m0 + row, k); }),
                 fn_loader([&](int row, int k) { return load_w(g, n0 + row, k); }), g.K, smem, acc);
    }
    tile_epilogue<EPI>(g, m0, n0, acc, smem);
}

// Persistent workgroups (3 per CU) streaming over tiles, buffer-load staging.  Logical workgroup
// id = xcd_remap(blockIdx): XCD x walks tiles [round*grid + x*grid/8, ...), i.e. whole row panels
// of A stay within one XCD's L2 in every round.
template <int EPI>
__global__ __launch_bounds__(THREADS, 3) void gemm_nt_f32_persistent_kernel(const GemmParams g, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles_n = (g.N + BN - 1) / BN;
    const int stride = gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= n_tiles) return;
    f32x4 ra[4], rw[4];
    {
        const BufferLoader la(g.A, g.M, g.lda, (tile / tiles_n) * BM), lw(g.W, g.N, g.ldw, (tile % tiles_n) * BN);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = la.load(i, 0);
            rw[i] = lw.load(i, 0);
        }
    }
    for (; tile < n_tiles; tile += stride) {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const int nt = tile + stride;
        const bool has_next = nt < n_tiles;
        const int nm0 = has_next ? (nt / tiles_n) * BM : m0, nn0 = has_next ? (nt % tiles_n) * BN : n0;
        f32x16 acc[2][2];
        mainloop_prefetched(BufferLoader(g.A, g.M, g.lda, m0), BufferLoader(g.W, g.N, g.ldw, n0),
                            BufferLoader(g.A, g.M, g.lda, nm0), BufferLoader(g.W, g.N, g.ldw, nn0), has_next, g.K, smem,
                            acc, ra, rw);
        tile_epilogue<EPI>(g, m0, n0, acc, smem);
    }
}

template <int EPI, int LOADER>
int launch(const GemmParams& g, hipStream_t stream) {
    hipLaunchKernelGGL((gemm_nt_f32_kernel<EPI, LOADER>), dim3(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN)),
                       dim3(THREADS), LDS_BYTES, stream, g);
    return pope_check_launch();
}

template <int EPI>
int launch_linear(const GemmParams& g, hipStream_t stream) {
    // fast path: K-steps never straddle a row end and every byte offset fits the 32-bit buffer offset
    const bool fast = (g.K % BK) == 0 && size_t(g.M + BM) * g.lda * 4 < (size_t(1) << 32) &&
                      size_t(g.N + BN) * g.ldw * 4 < (size_t(1) << 32);
    if (!fast) return launch<EPI, LOAD_GENERIC>(g, stream);
    const int n_tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const int slots = 3 * pope_cu_count();  // three resident workgroups per CU (LDS 36.9 KB, <=168 VGPRs)
    // Persistent streaming pays only when a slot sees few, short tiles (proj: 3 tiles of K=384 per
    // slot: +4 %); with more or longer tiles per slot the dispatcher's refill is 2-5 % faster
    // (lab A/B at M = 97 984: qkv 133 vs 131, fc1 137 vs 131, fc2 142 vs 139 TF/s).
    if (size_t(n_tiles) * g.K >= size_t(2048) * slots) return launch<EPI, LOAD_BUFFER>(g, stream);
    hipLaunchKernelGGL((gemm_nt_f32_persistent_kernel<EPI>), dim3(n_tiles < slots ? n_tiles : slots), dim3(THREADS),
                       LDS_BYTES, stream, g, n_tiles);
    return pope_check_launch();
}

}  // namespace

int pope_launch_gemm_nt_f32(const GemmParams& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K & 3) || (g.N & 3) || (g.ldc & 3)) return POPE_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(g.C) & 15) return POPE_ERR_ARG;
    if (g.epilogue != EPI_POSB && ((g.lda & 3) || (reinterpret_cast<uintptr_t>(g.A) & 15))) return POPE_ERR_ARG;
    if ((g.ldw & 3) || (reinterpret_cast<uintptr_t>(g.W) & 15)) return POPE_ERR_ARG;
    switch (g.epilogue) {
        case EPI_BIAS: return launch_linear<EPI_BIAS>(g, stream);
        case EPI_BIAS_GELU: return launch_linear<EPI_BIAS_GELU>(g, stream);
        case EPI_BIAS_LS_RES:
            if (!g.gamma || !g.res) return POPE_ERR_ARG;
            return launch_linear<EPI_BIAS_LS_RES>(g, stream);
        case EPI_POSB:
            if (!g.posb || g.ntok <= 0 || g.patch <= 0) return POPE_ERR_ARG;
            return launch<EPI_POSB, LOAD_PATCH>(g, stream);
        case EPI_CONV:   // the LoFTR stages' fp32 twin (conv.hip, loftr.hip, fine.hip): bias / shortcut / (leaky) ReLU, implicit 3x3
            if (g.res && (g.ldres & 3)) return POPE_ERR_ARG;
            if (g.conv_wp > 0) return (g.K % g.lda) ? POPE_ERR_ARG : launch<EPI_CONV, LOAD_CONV>(g, stream);
            return launch<EPI_CONV, LOAD_GENERIC>(g, stream);
    }
    return POPE_ERR_ARG;
}
