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

__device__ __forceinline__ float gelu_erf(float x) {
    // nn.GELU() default = exact erf form
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <bool PATCH>
__device__ __forceinline__ f32x4 load_a(const GemmParams& g, int row, int k) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= g.M || k >= g.K) return v;
    if constexpr (!PATCH) {
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

template <int EPI>
__device__ __forceinline__ void tile_epilogue(const GemmParams& g, int m0, int n0, const f32x16 (&acc)[2][2], float* smem) {
    // Coalesced epilogue (gemm_core::epilogue_rows): four consecutive output columns per call.
    epilogue_rows(acc, smem, [&](int tr, int tc, f32x4 v) {
        const int row = m0 + tr, col = n0 + tc;
        if (row >= g.M || col >= g.N) return;  // N % 4 == 0: a quad is inside or outside as a whole
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI != EPI_POSB)
            if (g.bias) bias = *reinterpret_cast<const f32x4*>(g.bias + col);
        if constexpr (EPI == EPI_BIAS) {
            v = v + bias;
        } else if constexpr (EPI == EPI_BIAS_GELU) {
            v = v + bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if constexpr (EPI == EPI_BIAS_LS_RES) {
            const f32x4 gamma = *reinterpret_cast<const f32x4*>(g.gamma + col);
            const f32x4 res = *reinterpret_cast<const f32x4*>(g.res + size_t(row) * g.ldres + col);
            v = res + (v + bias) * gamma;
        } else {  // EPI_POSB: + (pos_embed + conv bias | cls) table indexed by token
            v = v + *reinterpret_cast<const f32x4*>(g.posb + size_t(row % g.ntok) * g.N + col);
        }
        *reinterpret_cast<f32x4*>(g.C + size_t(row) * g.ldc + col) = v;
    });
}

enum { LOAD_GENERIC = 0, LOAD_BUFFER = 1, LOAD_PATCH = 2 };

// One tile per workgroup (generic pointer loader / im2col gather loader).
template <int EPI, int LOADER>
__global__ __launch_bounds__(THREADS, 3) void gemm_nt_f32_kernel(const GemmParams g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    f32x16 acc[2][2];
    mainloop(fn_loader([&](int row, int k) { return load_a<LOADER == LOAD_PATCH>(g, m0 + row, k); }),
             fn_loader([&](int row, int k) { return load_w(g, n0 + row, k); }), g.K, smem, acc);
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
    }
    return POPE_ERR_ARG;
}
