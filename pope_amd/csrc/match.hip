// Dense cross-image matcher: all-pairs similarity -> dual softmax -> threshold -> border
// removal -> mutual nearest neighbour -> ordered compaction -> index->pixel mapping.
//
// Semantics follow the reference CoarseMatching (src/matcher/utils/coarse_matching.py):
//   :109-114  f <- f / sqrt(C) (both sides), sim = f0 . f1^T / temperature
//   :119      conf = softmax(sim, dim=1) * softmax(sim, dim=2)
//   :175-184  mask = conf > thr ; zero a `border`-cell frame on all four grid dims (mask_border :8-25)
//   :187-189  mask &= (conf == rowmax(conf)) & (conf == colmax(conf))        (float equality)
//   :193-196  per row first True (mask.max(dim=2)); rows ordered by (b, i) (torch.where)
//   :242-250  mkpts = (idx % w, idx // w) * scale, (x, y) order, fp32
// The same kernels serve DINOv2 patch tokens (C=384, L=S=1530 at 476x630) and LoFTR coarse
// features (C=256).
//
// Round-1 structure: the contraction runs on the f32 MFMA (same 128x128 tile mainloop as the
// ViT GEMMs) and materialises sim once (the drop-in Matcher publishes conf_matrix anyway,
// matcher.py:71 / coarse_matching.py:145); the remaining passes are HBM/L2-bound streaming
// reductions, one wave per row or 64 columns per block, all coalesced.
#include "gemm_core.h"
#include "kernels.h"

namespace {

using namespace gemm_core;

__global__ __launch_bounds__(THREADS, 3) void sim_kernel(const MatchParams p, float inv_unused) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles_n = (p.S + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int pair = blockIdx.y;
    const float* f0 = p.feat0 + size_t(pair) * p.bs0;
    const float* f1 = p.feat1 + size_t(pair) * p.bs1;
    const float norm = sqrtf(float(p.C));  // feat / C**.5 as an fp32 division (coarse_matching.py:109)

    f32x16 acc[2][2];
    mainloop(fn_loader([&](int row, int k) {
                 f32x4 v = {0.f, 0.f, 0.f, 0.f};
                 if (m0 + row < p.L && k < p.C) v = *reinterpret_cast<const f32x4*>(f0 + size_t(m0 + row) * p.C + k) / norm;
                 return v;
             }),
             fn_loader([&](int row, int k) {
                 f32x4 v = {0.f, 0.f, 0.f, 0.f};
                 if (n0 + row < p.S && k < p.C) v = *reinterpret_cast<const f32x4*>(f1 + size_t(n0 + row) * p.C + k) / norm;
                 return v;
             }),
             p.C, smem, acc);

    float* sim = p.sim + size_t(pair) * p.L * p.S;
    const float temp = p.temperature;
    const bool even = !(p.S & 1);  // row starts are 8-byte aligned -> two float2 stores per quad
    epilogue_rows(acc, smem, [&](int tr, int tc, f32x4 v) {
        const int row = m0 + tr, col = n0 + tc;
        if (row >= p.L || col >= p.S) return;
        float* o = sim + size_t(row) * p.S + col;
        v = v / temp;
        if (even && col + 3 < p.S) {
            *reinterpret_cast<f32x2*>(o) = f32x2{v[0], v[1]};
            *reinterpret_cast<f32x2*>(o + 2) = f32x2{v[2], v[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < p.S) o[e] = v[e];
        }
    });
}

// softmax(sim, dim=2) statistics: one wave per row.
__global__ __launch_bounds__(256) void row_stats_kernel(const MatchParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), pair = blockIdx.y;
    if (row >= p.L) return;
    const float* x = p.sim + (size_t(pair) * p.L + row) * p.S;
    float m = -INFINITY;
    for (int s = lane; s < p.S; s += 64) m = fmaxf(m, x[s]);
    m = wave_max(m);
    float sum = 0.f;
    for (int s = lane; s < p.S; s += 64) sum += expf(x[s] - m);
    sum = wave_sum(sum);
    if (lane == 0) {
        p.row_max[size_t(pair) * p.L + row] = m;
        p.row_sum[size_t(pair) * p.L + row] = sum;
    }
}

// Column reductions: 64 columns per block, the 4 waves stride over the rows, combine via LDS.
// MODE 0: softmax(sim, dim=1) statistics (max, sum exp) ; MODE 1: column max of conf.
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const MatchParams p) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c, pair = blockIdx.y;
    const bool ok = col < p.S;
    const float* x = p.sim + size_t(pair) * p.L * p.S + col;
    float m = -INFINITY;
    if (ok)
        for (int l = g; l < p.L; l += 4) m = fmaxf(m, x[size_t(l) * p.S]);
    red[g][c] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][c], red[1][c]), fmaxf(red[2][c], red[3][c]));
    if constexpr (MODE == 1) {
        if (ok && g == 0) p.conf_colmax[size_t(pair) * p.S + col] = __float_as_uint(m);
        return;
    } else {
        __syncthreads();
        float sum = 0.f;
        if (ok)
            for (int l = g; l < p.L; l += 4) sum += expf(x[size_t(l) * p.S] - m);
        red[g][c] = sum;
        __syncthreads();
        if (ok && g == 0) {
            p.col_max[size_t(pair) * p.S + col] = m;
            p.col_sum[size_t(pair) * p.S + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        }
    }
}

// conf = softmax_dim1 * softmax_dim2, written in place over sim; row max of conf.
__global__ __launch_bounds__(256) void conf_kernel(const MatchParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), pair = blockIdx.y;
    if (row >= p.L) return;
    float* x = p.sim + (size_t(pair) * p.L + row) * p.S;
    const float rm = p.row_max[size_t(pair) * p.L + row], rsum = p.row_sum[size_t(pair) * p.L + row];
    const float* cm = p.col_max + size_t(pair) * p.S;
    const float* cs = p.col_sum + size_t(pair) * p.S;
    float best = 0.f;
    for (int s = lane; s < p.S; s += 64) {
        const float v = x[s];
        const float conf = (expf(v - cm[s]) / cs[s]) * (expf(v - rm) / rsum);
        x[s] = conf;
        best = fmaxf(best, conf);
    }
    best = wave_max(best);
    if (lane == 0) p.conf_rowmax[size_t(pair) * p.L + row] = best;
}

// Per row: first column passing threshold + border + mutual-NN equality tests.
__global__ __launch_bounds__(256) void select_kernel(const MatchParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), pair = blockIdx.y;
    if (row >= p.L) return;
    const int b = p.border > 0 ? p.border : 0;
    const int y0 = row / p.w0, x0 = row - y0 * p.w0;
    const bool row_ok = y0 >= b && y0 < p.h0 - b && x0 >= b && x0 < p.w0 - b;
    int first = 0x7fffffff;
    if (row_ok) {  // wave-uniform
        const float* x = p.sim + (size_t(pair) * p.L + row) * p.S;
        const float rmax = p.conf_rowmax[size_t(pair) * p.L + row];
        const unsigned* cmax = p.conf_colmax + size_t(pair) * p.S;
        for (int s0 = 0; s0 < p.S && first == 0x7fffffff; s0 += 64) {
            const int s = s0 + lane;
            bool hit = false;
            if (s < p.S) {
                const float conf = x[s];
                const int y1 = s / p.w1, x1 = s - y1 * p.w1;
                hit = conf > p.thr && conf == rmax && conf == __uint_as_float(cmax[s]) && y1 >= b &&
                      y1 < p.h1 - b && x1 >= b && x1 < p.w1 - b;
            }
            const unsigned long long ball = __ballot(hit);
            if (ball) first = s0 + __ffsll((long long)ball) - 1;
        }
    }
    if (lane == 0) {
        const size_t o = size_t(pair) * p.L + row;
        if (first != 0x7fffffff) {
            p.row_j[o] = first;
            p.row_conf[o] = p.sim[o * p.S + first];
        } else {
            p.row_j[o] = -1;
            p.row_conf[o] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void count_kernel(const MatchParams p) {
    __shared__ int wsum[4];
    const int pair = blockIdx.x;
    const int* rj = p.row_j + size_t(pair) * p.L;
    int c = 0;
    for (int l = threadIdx.x; l < p.L; l += 256) c += rj[l] >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) p.counts[pair] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Ordered compaction: block per pair; rows ascending -> output ordered by (b, i) like torch.where.
__global__ __launch_bounds__(256) void scatter_kernel(const MatchParams p) {
    __shared__ int wcnt[4];
    __shared__ int base_s;
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        int base = 0;
        for (int i = 0; i < pair; ++i) base += p.counts[i];
        base_s = base;
        if (pair == p.n - 1) p.counts[p.n] = base + p.counts[pair];
    }
    __syncthreads();
    int base = base_s;
    const int* rj = p.row_j + size_t(pair) * p.L;
    const float* rc = p.row_conf + size_t(pair) * p.L;
    for (int l0 = 0; l0 < p.L; l0 += 256) {
        const int l = l0 + threadIdx.x;
        const int j = l < p.L ? rj[l] : -1;
        const bool hit = j >= 0;
        const unsigned long long ball = __ballot(hit);
        if (lane == 0) wcnt[wave] = __popcll(ball);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (hit) {
            const int o = off + __popcll(ball & ((1ull << lane) - 1ull));
            p.b_ids[o] = pair;
            p.i_ids[o] = l;
            p.j_ids[o] = j;
            p.mconf[o] = rc[l];
            p.mkpts0[2 * o] = float(l % p.w0) * p.scale;
            p.mkpts0[2 * o + 1] = float(l / p.w0) * p.scale;
            p.mkpts1[2 * o] = float(j % p.w1) * p.scale;
            p.mkpts1[2 * o + 1] = float(j / p.w1) * p.scale;
        }
        base += total;
        __syncthreads();
    }
}

}  // namespace

int pope_launch_dense_match_f32(const MatchParams& p, hipStream_t stream) {
    if (p.n <= 0 || p.L <= 0 || p.S <= 0 || p.C <= 0 || (p.C & 3) || p.n > 65535) return POPE_ERR_ARG;
    if (p.L != p.h0 * p.w0 || p.S != p.h1 * p.w1) return POPE_ERR_ARG;
    if (p.bs0 < (long long)p.L * p.C || p.bs1 < (long long)p.S * p.C || (p.bs0 & 3) || (p.bs1 & 3)) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p.feat0) & 15) || (reinterpret_cast<uintptr_t>(p.feat1) & 15)) return POPE_ERR_ARG;
    bool sim_done = false;
    if (p.planes0 && p.planes1 && (p.C & 31) == 0 && p.C >= 64) {
        // f16x3: (f0 / sqrt(C)) and (f1 / sqrt(C)) as hi/lo planes (x256), one batched planes GEMM, (acc / 2^16) / T
        const float norm = sqrtf(float(p.C));
        int rc = pope_launch_div_planes(p.feat0, p.bs0, p.planes0, p.n, p.L, p.C, norm, K_PLANES_W_SCALE, p.range_flag, stream);
        if (!rc) rc = pope_launch_div_planes(p.feat1, p.bs1, p.planes1, p.n, p.S, p.C, norm, K_PLANES_W_SCALE, p.range_flag, stream);
        if (rc) return rc;
        GemmParams g = {};
        g.a_pl = p.planes0; g.w_pl = p.planes1; g.C = p.sim;
        g.M = p.L; g.N = p.S; g.K = p.C; g.lda = p.C; g.ldw = p.C; g.ldc = p.S;
        g.nbatch = p.n;
        g.alpha = 1.0f / (K_PLANES_W_SCALE * K_PLANES_W_SCALE);
        g.divisor = p.temperature;
        g.epilogue = EPI_SIM;
        rc = pope_launch_sim_f16x3_planes(g, stream);
        if (rc == 0) sim_done = true;
        else if (rc != POPE_ERR_ARG) return rc;  // shapes beyond the 32-bit offsets: the fp32 kernel below
    }
    if (!sim_done) {
        const int tiles = ((p.L + BM - 1) / BM) * ((p.S + BN - 1) / BN);
        hipLaunchKernelGGL(sim_kernel, dim3(tiles, p.n), dim3(THREADS), LDS_BYTES, stream, p, 0.f);
    }
    const dim3 rows((p.L + 3) / 4, p.n), cols((p.S + 63) / 64, p.n);
    hipLaunchKernelGGL(row_stats_kernel, rows, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(col_reduce_kernel<0>, cols, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(conf_kernel, rows, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(col_reduce_kernel<1>, cols, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(select_kernel, rows, dim3(256), 0, stream, p);
    hipLaunchKernelGGL(count_kernel, dim3(p.n), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(scatter_kernel, dim3(p.n), dim3(256), 0, stream, p);
    return pope_check_launch();
}
