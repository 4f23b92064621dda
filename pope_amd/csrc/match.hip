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
// Structure (round 2): the L x S matrix is written ONCE (sim, by the contraction) and read ONCE:
//   1. contraction on the matrix cores (f16x3 planes GEMM, gemm_planes.hip EPI_SIM) whose epilogue also emits the
//      per-tile pieces of the row and column softmax statistics from its registers; `combine_*_kernel` folds the
//      pieces (KBs per pair).  [fp32 mode: sim_kernel on the fp32 MFMA + two streaming statistics passes.]
//   2. conf_pass_kernel: one streaming pass that forms conf = softmax_dim1 * softmax_dim2, optionally publishes it in
//      place (the drop-in CoarseMatching does; the batch pipeline does not and never writes the matrix again), and
//      keeps per row (max, first argmax, number of argmax ties) and per 32-row block the column maxima.
//   3. select_kernel: per row a constant-time test of its argmax column against the column maxima (rows with tied
//      maxima are re-scanned: the reference takes the first column that passes ALL tests), then the ordered
//      compaction.
// The reference makes >= 5 passes over the matrix; round 1 made 8.
#include "gemm_core.h"
#include "kernels.h"

namespace {

using namespace gemm_core;

constexpr float L2E = 1.44269504088896340736f;

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

// ---- softmax statistics, fp32-mode path: two streaming passes over sim ---------------------------------------------
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
    for (int s = lane; s < p.S; s += 64) sum += __builtin_amdgcn_exp2f((x[s] - m) * L2E);
    sum = wave_sum(sum);
    if (lane == 0) {
        p.row_max[size_t(pair) * p.L + row] = m;
        p.row_sum[size_t(pair) * p.L + row] = sum;
    }
}

// softmax(sim, dim=1) statistics: 64 columns per block, the 4 waves stride over the rows, combine via LDS.
__global__ __launch_bounds__(256) void col_stats_kernel(const MatchParams p) {
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
    __syncthreads();
    float sum = 0.f;
    if (ok)
        for (int l = g; l < p.L; l += 4) sum += __builtin_amdgcn_exp2f((x[size_t(l) * p.S] - m) * L2E);
    red[g][c] = sum;
    __syncthreads();
    if (ok && g == 0) {
        p.col_max[size_t(pair) * p.S + col] = m;
        p.col_sum[size_t(pair) * p.S + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// ---- softmax statistics, f16x3 path: fold the per-tile pieces the contraction's epilogue wrote ----------------------
// A piece is (max, sum of exp(. - max)) over a block of columns (rows); pieces fold like the online softmax.  Fixed
// order over the pieces: identical rows (columns) get identical statistics whatever their position.
__global__ __launch_bounds__(256) void combine_row_stats_kernel(const MatchParams p) {
    const size_t row = size_t(blockIdx.x) * 256 + threadIdx.x;   // over n * L
    if (row >= size_t(p.n) * p.L) return;
    const f32x2* part = reinterpret_cast<const f32x2*>(p.row_part) + row * p.ncb;
    float m = -INFINITY;
    for (int b = 0; b < p.ncb; ++b) m = fmaxf(m, part[b][0]);
    float sum = 0.f;
    for (int b = 0; b < p.ncb; ++b) {
        const f32x2 v = part[b];
        if (v[0] != -INFINITY) sum += v[1] * __builtin_amdgcn_exp2f((v[0] - m) * L2E);
    }
    p.row_max[row] = m;
    p.row_sum[row] = sum;
}

__global__ __launch_bounds__(256) void combine_col_stats_kernel(const MatchParams p) {
    const int col = blockIdx.x * 256 + threadIdx.x, pair = blockIdx.y;
    if (col >= p.S) return;
    const float* pm = p.col_pmax + size_t(pair) * p.nrb * p.ldp + col;
    const float* ps = p.col_psum + size_t(pair) * p.nrb * p.ldp + col;
    float m = -INFINITY;
    for (int b = 0; b < p.nrb; ++b) m = fmaxf(m, pm[size_t(b) * p.ldp]);
    float sum = 0.f;
    for (int b = 0; b < p.nrb; ++b) {
        const float v = pm[size_t(b) * p.ldp];
        if (v != -INFINITY) sum += ps[size_t(b) * p.ldp] * __builtin_amdgcn_exp2f((v - m) * L2E);
    }
    p.col_max[size_t(pair) * p.S + col] = m;
    p.col_sum[size_t(pair) * p.S + col] = sum;
}

// ---- the one pass over the matrix -----------------------------------------------------------------------------------
// conf(i, j) from sim and the four statistics: softmax(sim, dim=1) * softmax(sim, dim=2) (coarse_matching.py:119, in
// that order).  ONE definition: conf_pass_kernel and the tie re-scan of select_kernel must produce the same bits.
// exp(v - cmx) / csum * exp(v - rmx) / rsum with ONE exponential: exp((v - cmx) + (v - rmx)) * (cinv * rinv).  Both
// differences are <= 0 and computed first (exact where it matters: near the maxima), so the exponent is as accurate as
// in the two-exponential form; a product that underflowed there underflows here.  The pair form is the same arithmetic
// in packed instructions (v_pk_add / v_pk_mul are IEEE-identical to the scalar ones; contraction is off).
__device__ __forceinline__ float conf_value(float v, float cmx, float cinv, float rmx, float rinv) {
    return __builtin_amdgcn_exp2f(((v - cmx) + (v - rmx)) * L2E) * (cinv * rinv);
}
__device__ __forceinline__ f32x2 conf_value2(f32x2 v, f32x2 cmx, f32x2 cinv, float rmx, float rinv) {
    const f32x2 t = ((v - cmx) + (v - f32x2{rmx, rmx})) * f32x2{L2E, L2E};
    return f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} * (cinv * f32x2{rinv, rinv});
}

constexpr int CP_ROWS = 32;   // rows per workgroup: 8 per wave
constexpr int CP_K = 16;      // column pairs per lane and chunk: a chunk is 64 * 2 * 16 = 2048 columns
constexpr int CP_CHUNK = 64 * 2 * CP_K;

struct RowBest {   // running (max, first argmax, number of argmax ties) of a row
    float v;
    int idx, cnt;
    __device__ __forceinline__ void take(float c, int col) {
        const bool gt = c > v, eq = c == v;
        cnt = gt ? 1 : cnt + (eq ? 1 : 0);
        idx = gt ? col : idx;   // columns arrive in ascending order per lane: a tie keeps the earlier column
        v = gt ? c : v;
    }
    __device__ __forceinline__ void merge(float ov, int oidx, int ocnt) {
        const bool gt = ov > v, eq = ov == v;
        cnt = gt ? ocnt : cnt + (eq ? ocnt : 0);
        idx = gt ? oidx : (eq && oidx < idx ? oidx : idx);
        v = gt ? ov : v;
    }
};

// Workgroup = 32 rows of one pair x all columns.  Lane l of every wave owns columns c0 + 128 k + 2 l (+1): it keeps
// their column statistics and their running column maximum in registers while the wave walks its 8 rows, so a row is
// read as 8-byte pieces that make whole 512-byte wave transactions.  VEC2 = false (odd S: rows are only 4-byte
// aligned): the same with single columns c0 + 64 k + l and half the chunk.
template <bool PUBLISH, bool VEC2>
__global__ __launch_bounds__(256) void conf_pass_kernel(const MatchParams p) {
    __shared__ float cmax_s[4][CP_CHUNK];
    constexpr int W = VEC2 ? 2 : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.y, blk = blockIdx.x;
    const int row0 = blk * CP_ROWS + wave * (CP_ROWS / 4);
    const size_t prow = size_t(pair) * p.L;
    const float* cmx_g = p.col_max + size_t(pair) * p.S;
    const float* csum_g = p.col_sum + size_t(pair) * p.S;
    RowBest best[CP_ROWS / 4];
#pragma unroll
    for (int r = 0; r < CP_ROWS / 4; ++r) best[r] = RowBest{-1.f, 0, 0};   // conf >= 0

    for (int c0 = 0; c0 < p.S; c0 += 64 * W * CP_K) {
        float cmx[CP_K][W], cinv[CP_K][W], cbest[CP_K][W];
#pragma unroll
        for (int k = 0; k < CP_K; ++k)
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const int col = c0 + 64 * W * k + W * lane + e;
                const bool ok = col < p.S;
                cmx[k][e] = ok ? cmx_g[col] : 0.f;
                cinv[k][e] = ok ? 1.0f / csum_g[col] : 0.f;
                cbest[k][e] = 0.f;
            }
#pragma unroll
        for (int r = 0; r < CP_ROWS / 4; ++r) {
            const int row = row0 + r;
            if (row >= p.L) break;   // wave-uniform
            const float rmx = p.row_max[prow + row], rinv = 1.0f / p.row_sum[prow + row];
            float* x = p.sim + (prow + row) * p.S;
#pragma unroll
            for (int k = 0; k < CP_K; ++k) {
                const int col = c0 + 64 * W * k + W * lane;
                if (c0 + 64 * W * k >= p.S) break;   // wave-uniform
                if constexpr (VEC2) {
                    if (col + 1 < p.S) {
                        const f32x2 v = *reinterpret_cast<const f32x2*>(x + col);
                        const f32x2 c = {conf_value(v[0], cmx[k][0], cinv[k][0], rmx, rinv),
                                         conf_value(v[1], cmx[k][1], cinv[k][1], rmx, rinv)};
                        if constexpr (PUBLISH) *reinterpret_cast<f32x2*>(x + col) = c;
                        best[r].take(c[0], col);
                        best[r].take(c[1], col + 1);
                        cbest[k][0] = fmaxf(cbest[k][0], c[0]);
                        cbest[k][1] = fmaxf(cbest[k][1], c[1]);
                    } else if (col < p.S) {
                        const float c = conf_value(x[col], cmx[k][0], cinv[k][0], rmx, rinv);
                        if constexpr (PUBLISH) x[col] = c;
                        best[r].take(c, col);
                        cbest[k][0] = fmaxf(cbest[k][0], c);
                    }
                } else if (col < p.S) {
                    const float c = conf_value(x[col], cmx[k][0], cinv[k][0], rmx, rinv);
                    if constexpr (PUBLISH) x[col] = c;
                    best[r].take(c, col);
                    cbest[k][0] = fmaxf(cbest[k][0], c);
                }
            }
        }
        // column maxima of this 32-row block: the four waves' registers meet in LDS
        if (c0) __syncthreads();
#pragma unroll
        for (int k = 0; k < CP_K; ++k)
#pragma unroll
            for (int e = 0; e < W; ++e) cmax_s[wave][64 * W * k + W * lane + e] = cbest[k][e];
        __syncthreads();
        float* out = p.colmax_part + (size_t(pair) * p.nrb2 + blk) * p.ldp;
        for (int c = threadIdx.x; c < 64 * W * CP_K && c0 + c < p.S; c += 256)
            out[c0 + c] = fmaxf(fmaxf(cmax_s[0][c], cmax_s[1][c]), fmaxf(cmax_s[2][c], cmax_s[3][c]));
    }
    // per row: reduce the lanes' candidates (first argmax = smallest column among the maxima)
#pragma unroll
    for (int r = 0; r < CP_ROWS / 4; ++r) {
        const int row = row0 + r;
        if (row >= p.L) break;
        RowBest b = best[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b.merge(__shfl_xor(b.v, o), __shfl_xor(b.idx, o), __shfl_xor(b.cnt, o));
        if (lane == 0) {
            p.conf_rowmax[prow + row] = b.v;
            p.row_arg[prow + row] = b.idx;
            p.row_cnt[prow + row] = b.cnt;
        }
    }
}

__global__ __launch_bounds__(256) void combine_colmax_kernel(const MatchParams p) {
    const int col = blockIdx.x * 256 + threadIdx.x, pair = blockIdx.y;
    if (col >= p.S) return;
    const float* part = p.colmax_part + size_t(pair) * p.nrb2 * p.ldp + col;
    float m = 0.f;
    for (int b = 0; b < p.nrb2; ++b) m = fmaxf(m, part[size_t(b) * p.ldp]);
    p.conf_colmax[size_t(pair) * p.S + col] = m;
}

// Per row (one wave): the first column passing threshold + border + mutual-NN equality tests.  Only a column that
// attains the row maximum can pass; with a unique argmax (count == 1: practically always) that is one test, otherwise
// the row is re-scanned for the first argmax column that passes all tests, as `mask.max(dim=2)` does.
template <bool PUBLISHED>
__global__ __launch_bounds__(256) void select_kernel(const MatchParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), pair = blockIdx.y;
    if (row >= p.L) return;
    const int b = p.border > 0 ? p.border : 0;
    const int y0 = row / p.w0, x0 = row - y0 * p.w0;
    const size_t o = size_t(pair) * p.L + row;
    const float rmax = p.conf_rowmax[o];
    const bool row_ok = y0 >= b && y0 < p.h0 - b && x0 >= b && x0 < p.w0 - b && rmax > p.thr;
    const float* cmax = p.conf_colmax + size_t(pair) * p.S;
    auto col_ok = [&](int s) {
        const int y1 = s / p.w1, x1 = s - y1 * p.w1;
        return y1 >= b && y1 < p.h1 - b && x1 >= b && x1 < p.w1 - b && rmax == cmax[s];
    };
    int first = -1;
    if (row_ok) {  // wave-uniform
        if (p.row_cnt[o] == 1) {
            const int s = p.row_arg[o];
            if (col_ok(s)) first = s;
        } else {
            const float* x = p.sim + o * p.S;
            const float rmx = p.row_max[o], rinv = 1.0f / p.row_sum[o];
            const float* cmx = p.col_max + size_t(pair) * p.S;
            const float* csum = p.col_sum + size_t(pair) * p.S;
            for (int s0 = 0; s0 < p.S && first < 0; s0 += 64) {
                const int s = s0 + lane;
                bool hit = false;
                if (s < p.S) {
                    const float conf = PUBLISHED ? x[s] : conf_value(x[s], cmx[s], 1.0f / csum[s], rmx, rinv);
                    hit = conf == rmax && col_ok(s);
                }
                const unsigned long long ball = __ballot(hit);
                if (ball) first = s0 + __ffsll((long long)ball) - 1;
            }
        }
    }
    if (lane == 0) {
        p.row_j[o] = first;
        p.row_conf[o] = first >= 0 ? rmax : 0.f;
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

// Fast form for even S (rows 8-byte aligned): no control flow inside a row.  A lane owns the column pairs
// c0 + 128 k + 2 l (k < NK); a row's NK 8-byte pieces are fetched by bounds-checked buffer loads (columns past S read
// 0 and give conf 0, their stores are dropped), all in flight together and one row ahead of the arithmetic; conf, the
// running column maxima and the row maximum cost two max + one compare per element; the first argmax and the tie
// count come from wave ballots on the scalar unit.  NK = ceil(S / 128) for S <= 2048 (DINOv2 tokens at 476 x 630: 12;
// LoFTR 256^2: 8); wider matrices are walked in chunks of 2048 columns, lane 0 merging a row's chunk results.
template <bool PUBLISH, int NK>
__global__ __launch_bounds__(256) void conf_pass_fast_kernel(const MatchParams p) {
    __shared__ float cmax_s[4][128 * NK];
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2m;
    constexpr int RPW = CP_ROWS / 4;   // rows per wave
    constexpr int BIG = 1 << 24;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.y, blk = blockIdx.x;
    const int row0 = blk * CP_ROWS + wave * RPW;
    const size_t prow = size_t(pair) * p.L;
    const float* cmx_g = p.col_max + size_t(pair) * p.S;
    const float* csum_g = p.col_sum + size_t(pair) * p.S;
    const int last_row = p.L - 1;

    for (int c0 = 0; c0 < p.S; c0 += 128 * NK) {
        f32x2 cmx[NK], cinv[NK], cbest[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int col = c0 + 128 * k + 2 * lane + e;
                const bool ok = col < p.S;
                cmx[k][e] = ok ? cmx_g[col] : 0.f;
                cinv[k][e] = ok ? 1.0f / csum_g[col] : 0.f;
                cbest[k][e] = 0.f;
            }
        const unsigned voff = unsigned(c0 + 2 * lane) * 4u;
        auto row_rsrc = [&](int row) {   // rows past the end re-read the last row (never used)
            const int rr = row < last_row ? row : last_row;
            return __builtin_amdgcn_make_buffer_rsrc(p.sim + (prow + rr) * p.S, 0, unsigned(p.S) * 4u, 0x00020000);
        };
        auto fetch = [&](int row, f32x2 (&c)[NK]) {
            const __amdgpu_buffer_rsrc_t rs = row_rsrc(row);
#pragma unroll
            for (int k = 0; k < NK; ++k) c[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 512 * k, 0));
        };
        auto process = [&](int row, f32x2 (&c)[NK]) {
            if (row > last_row) return;   // wave-uniform, tail block only
            const float rmx = p.row_max[prow + row], rinv = 1.0f / p.row_sum[prow + row];
            float m = 0.f;   // conf >= 0
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                c[k] = conf_value2(c[k], cmx[k], cinv[k], rmx, rinv);
                cbest[k][0] = fmaxf(cbest[k][0], c[k][0]);
                cbest[k][1] = fmaxf(cbest[k][1], c[k][1]);
                m = fmaxf(m, fmaxf(c[k][0], c[k][1]));
            }
            if constexpr (PUBLISH) {
                const __amdgpu_buffer_rsrc_t rs = row_rsrc(row);
#pragma unroll
                for (int k = 0; k < NK; ++k) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2m, c[k]), rs, voff, 512 * k, 0);
            }
            m = wave_max(m);
            // first argmax and tie count within this chunk, on the scalar unit (columns ascend with k, then lane, then e;
            // asm selects: the compiler must not turn this into 2 NK branches)
            int first = BIG, cnt = 0;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const unsigned long long m0 = __ballot(c[k][0] == m), m1 = __ballot(c[k][1] == m);
                cnt += __popcll(m0) + __popcll(m1);
                const int a = pope_uniform_select(m0 != 0, 2 * __builtin_ctzll(m0 | (1ull << 63)), BIG);
                const int b2 = pope_uniform_select(m1 != 0, 2 * __builtin_ctzll(m1 | (1ull << 63)) + 1, BIG);
                const int cand = c0 + 128 * k + (a < b2 ? a : b2);
                first = first < cand ? first : cand;   // BIG + offsets stay far above any column
            }
            if (lane == 0) {
                float bv = m;
                int bi = first, bc = cnt;
                if (c0) {   // merge with the earlier chunks of this row (same lane wrote them)
                    const float pv = p.conf_rowmax[prow + row];
                    const int pi = p.row_arg[prow + row], pc = p.row_cnt[prow + row];
                    if (pv > m) { bv = pv; bi = pi; bc = pc; }
                    else if (pv == m) { bi = pi; bc = pc + cnt; }
                }
                p.conf_rowmax[prow + row] = bv;
                p.row_arg[prow + row] = bi;
                p.row_cnt[prow + row] = bc;
            }
        };
        f32x2 ca[NK], cb[NK];
        fetch(row0, ca);
#pragma unroll 1
        for (int r = 0; r < RPW; r += 2) {
            fetch(row0 + r + 1, cb);
            process(row0 + r, ca);
            fetch(row0 + r + 2, ca);
            process(row0 + r + 1, cb);
        }
        if (c0) __syncthreads();
#pragma unroll
        for (int k = 0; k < NK; ++k) *reinterpret_cast<f32x2*>(&cmax_s[wave][128 * k + 2 * lane]) = cbest[k];
        __syncthreads();
        float* out = p.colmax_part + (size_t(pair) * p.nrb2 + blk) * p.ldp;
        for (int cc = threadIdx.x; cc < 128 * NK && c0 + cc < p.S; cc += 256)
            out[c0 + cc] = fmaxf(fmaxf(cmax_s[0][cc], cmax_s[1][cc]), fmaxf(cmax_s[2][cc], cmax_s[3][cc]));
    }
}

template <bool PUBLISH>
void launch_conf_pass(const MatchParams& p, hipStream_t stream) {
    const dim3 blocks((p.L + CP_ROWS - 1) / CP_ROWS, p.n);
    if (p.S & 1) {   // odd S: rows are only 4-byte aligned
        hipLaunchKernelGGL((conf_pass_kernel<PUBLISH, false>), blocks, dim3(256), 0, stream, p);
        return;
    }
    const int nk = (p.S + 127) / 128;
    if (nk <= 4) hipLaunchKernelGGL((conf_pass_fast_kernel<PUBLISH, 4>), blocks, dim3(256), 0, stream, p);
    else if (nk <= 8) hipLaunchKernelGGL((conf_pass_fast_kernel<PUBLISH, 8>), blocks, dim3(256), 0, stream, p);
    else if (nk <= 12) hipLaunchKernelGGL((conf_pass_fast_kernel<PUBLISH, 12>), blocks, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conf_pass_fast_kernel<PUBLISH, 16>), blocks, dim3(256), 0, stream, p);
}

template <bool PUBLISH>
void launch_conf_and_select(const MatchParams& p, hipStream_t stream) {
    launch_conf_pass<PUBLISH>(p, stream);
    hipLaunchKernelGGL(combine_colmax_kernel, dim3((p.S + 255) / 256, p.n), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(select_kernel<PUBLISH>, dim3((p.L + 3) / 4, p.n), dim3(256), 0, stream, p);
}

}  // namespace

int pope_match_nrb2(int L) { return (L + CP_ROWS - 1) / CP_ROWS; }

int pope_launch_dense_match_f32(const MatchParams& p, hipStream_t stream) {
    if (p.n <= 0 || p.L <= 0 || p.S <= 0 || p.C <= 0 || (p.C & 3) || p.n > 65535) return POPE_ERR_ARG;
    if (p.L != p.h0 * p.w0 || p.S != p.h1 * p.w1) return POPE_ERR_ARG;
    if (p.bs0 < (long long)p.L * p.C || p.bs1 < (long long)p.S * p.C || (p.bs0 & 3) || (p.bs1 & 3)) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p.feat0) & 15) || (reinterpret_cast<uintptr_t>(p.feat1) & 15)) return POPE_ERR_ARG;
    if (!p.sim || !p.colmax_part || p.nrb2 != pope_match_nrb2(p.L) || p.ldp < p.S || (p.ldp & 3)) return POPE_ERR_ARG;
    bool sim_done = false;
    if (p.planes0 && p.planes1 && p.row_part && p.col_pmax && p.col_psum && (p.C & 31) == 0 && p.C >= 64) {
        // f16x3: (f0 / sqrt(C)) and (f1 / sqrt(C)) as hi/lo planes (x256), one batched planes GEMM whose epilogue
        // divides by T * 2^16 and emits the pieces of the softmax statistics
        const float norm = sqrtf(float(p.C));
        int rc = pope_launch_div_planes(p.feat0, p.bs0, p.planes0, p.n, p.L, p.C, norm, K_PLANES_W_SCALE, p.range_flag, stream);
        if (!rc) rc = pope_launch_div_planes(p.feat1, p.bs1, p.planes1, p.n, p.S, p.C, norm, K_PLANES_W_SCALE, p.range_flag, stream);
        if (rc) return rc;
        GemmParams g = {};
        g.a_pl = p.planes0; g.w_pl = p.planes1; g.C = p.sim;
        g.M = p.L; g.N = p.S; g.K = p.C; g.lda = p.C; g.ldw = p.C; g.ldc = p.S;
        g.nbatch = p.n;
        g.divisor_eff = p.temperature * (K_PLANES_W_SCALE * K_PLANES_W_SCALE);   // exact: the scales are powers of two
        g.rdiv = 1.0f / g.divisor_eff;
        g.epilogue = EPI_SIM;
        g.row_part = p.row_part; g.col_pmax = p.col_pmax; g.col_psum = p.col_psum;
        g.ncb = p.ncb; g.nrb = p.nrb; g.ldp = p.ldp;
        rc = pope_launch_sim_f16x3_planes(g, stream);
        if (rc == 0) {
            sim_done = true;
            hipLaunchKernelGGL(combine_row_stats_kernel, dim3(unsigned((size_t(p.n) * p.L + 255) / 256)), dim3(256), 0, stream, p);
            hipLaunchKernelGGL(combine_col_stats_kernel, dim3((p.S + 255) / 256, p.n), dim3(256), 0, stream, p);
        } else if (rc != POPE_ERR_ARG) {
            return rc;  // shapes beyond the 32-bit offsets: the fp32 kernels below
        }
    }
    if (!sim_done) {
        const int tiles = ((p.L + BM - 1) / BM) * ((p.S + BN - 1) / BN);
        hipLaunchKernelGGL(sim_kernel, dim3(tiles, p.n), dim3(THREADS), LDS_BYTES, stream, p, 0.f);
        hipLaunchKernelGGL(row_stats_kernel, dim3((p.L + 3) / 4, p.n), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(col_stats_kernel, dim3((p.S + 63) / 64, p.n), dim3(256), 0, stream, p);
    }
    if (p.publish_conf) launch_conf_and_select<true>(p, stream);
    else launch_conf_and_select<false>(p, stream);
    hipLaunchKernelGGL(count_kernel, dim3(p.n), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(scatter_kernel, dim3(p.n), dim3(256), 0, stream, p);
    return pope_check_launch();
}
