// LoFTR fine stage around its two transformer layers:
//   FinePreprocess.forward  src/matcher/loftr_module/fine_preprocess.py:29-59 — W x W windows (W = 5, zero padded by W/2)
//     of the 1/2-resolution maps at the matched coarse cells, the matched coarse features through down_proj, both
//     through merge_feat.  The reference unfolds EVERY window of the map and indexes afterwards ([n, L, 25, 128]); here
//     only the M matched windows are gathered, straight into the planes rows the merge GEMM reads.
//   FineMatching.forward / get_fine_match  src/matcher/utils/fine_matching.py:15-74 — centre-vs-window correlation,
//     softmax (temperature 1 / sqrt(C)), expectation and standard deviation over the normalised [-1, 1]^2 grid,
//     mkpts1_f = mkpts1_c + coords * (W / 2) * scale.
// The two Linear layers run on the f16x3 planes GEMM (gemm_planes.hip); everything else is one gather or reduce kernel.
#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr float A_SCALE = K_PLANES_ACT_SCALE;

template <bool PLANES = true>   // false (the fp32 twin): `row` is a row of floats of the same byte pitch
__device__ __forceinline__ void store_planes4(_Float16* row, int c, f32x4 v, float& amax) {
    if constexpr (!PLANES) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(row) + c) = v;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        amax = fmaxf(amax, fabsf(v[e]));
        if (!(v[e] == v[e])) amax = INFINITY;
    }
    f16x4 hi, lo;
    pope_split4(v * A_SCALE, hi, lo);
    _Float16* o = row + (c >> 5) * 64 + (c & 31);
    *reinterpret_cast<f16x4*>(o) = hi;
    *reinterpret_cast<f16x4*>(o + 32) = lo;
}

// rows [2M, Cc] of the matched coarse features (feat_c0[b, i] then feat_c1[b, j]) as activation planes
template <bool PLANES>
__global__ __launch_bounds__(256) void fine_gather_coarse_kernel(const float* __restrict__ fc0, const float* __restrict__ fc1,
                                                                 const long long* __restrict__ b_ids, const long long* __restrict__ i_ids,
                                                                 const long long* __restrict__ j_ids, int M, int L, int S, int Cc,
                                                                 _Float16* __restrict__ out, unsigned* range_flag) {
    const int groups = Cc / 4;
    const long long total = 2ll * M * groups;
    float amax = 0.f;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int g = int(id % groups);
        const int r = int(id / groups), m = r < M ? r : r - M;
        const long long b = b_ids[m];
        const float* src = r < M ? fc0 + (b * L + i_ids[m]) * Cc : fc1 + (b * S + j_ids[m]) * Cc;
        store_planes4<PLANES>(out + (size_t)r * 2 * Cc, 4 * g, *reinterpret_cast<const f32x4*>(src + 4 * g), amax);
    }
    if constexpr (PLANES) pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax * A_SCALE < POPE_F16_OVERFLOW));
}

// rows [2M * WW, 2 Cf] of the merge_feat input as planes: columns [0, Cf) = window position k of match m in the fine
// map of its stream (zero outside the map), columns [Cf, 2 Cf) = the match's down-projected coarse feature.  The fine
// maps are addressed through element strides (NCHW tensors and NHWC views alike).
struct FineMap { const float* p; long long sn, sc, sh, sw; int H, W, wc; };
template <bool PLANES>
__global__ __launch_bounds__(256) void fine_gather_windows_kernel(FineMap f0, FineMap f1, const float* __restrict__ c_win,
                                                                  const long long* __restrict__ b_ids, const long long* __restrict__ i_ids,
                                                                  const long long* __restrict__ j_ids, int M, int Wn, int stride, int Cf,
                                                                  _Float16* __restrict__ out, unsigned* range_flag) {
    const int WW = Wn * Wn, groups = 2 * Cf / 4, pad = Wn / 2;
    const long long total = 2ll * M * WW * groups;
    float amax = 0.f;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int g = int(id % groups);
        const long long row = id / groups;
        const int k = int(row % WW), r = int(row / WW), m = r < M ? r : r - M;
        const int c = 4 * g;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c >= Cf) {
            v = *reinterpret_cast<const f32x4*>(c_win + (size_t)r * Cf + (c - Cf));
        } else {
            const FineMap& f = r < M ? f0 : f1;
            const long long cell = r < M ? i_ids[m] : j_ids[m];
            const int y = int(cell / f.wc) * stride + k / Wn - pad, x = int(cell % f.wc) * stride + k % Wn - pad;
            if (y >= 0 && y < f.H && x >= 0 && x < f.W) {
                const float* s = f.p + b_ids[m] * f.sn + y * f.sh + x * f.sw + c * f.sc;
                if (f.sc == 1) v = *reinterpret_cast<const f32x4*>(s);
                else v = f32x4{s[0], s[f.sc], s[2 * f.sc], s[3 * f.sc]};
            }
        }
        store_planes4<PLANES>(out + (size_t)row * 4 * Cf, c, v, amax);
    }
    if constexpr (PLANES) pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax * A_SCALE < POPE_F16_OVERFLOW));
}

// one wave per match: sim[k] = <win0[m, centre], win1[m, k]> / sqrt(C), softmax over the WW positions, expectation
// of (x, y) on linspace(-1, 1, W)^2, spread = sum over the two axes of sqrt(max(E[g^2] - E[g]^2, 1e-10))
__global__ __launch_bounds__(256) void fine_match_kernel(const float* __restrict__ win0, const float* __restrict__ win1, int M, int Wn,
                                                         int C, const float* __restrict__ mkpts1_c, float scale_px,
                                                         float* __restrict__ expec, float* __restrict__ mkpts1_f) {
    const int WW = Wn * Wn;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* centre = win0 + ((size_t)m * WW + WW / 2) * C;
    const float temp = 1.0f / sqrtf(float(C));
    // lane k < WW owns window position k (WW <= 64)
    float s = -INFINITY;
    if (lane < WW) {
        const float* w = win1 + ((size_t)m * WW + lane) * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += centre[c] * w[c];   // the einsum's contraction, c ascending
        s = acc * temp;
    }
    const float mx = wave_max(s);
    const float e = lane < WW ? expf(s - mx) : 0.f;
    const float p = e / wave_sum(e);
    const float step = Wn > 1 ? 2.0f / float(Wn - 1) : 0.f;
    const float gx = lane < WW ? -1.0f + step * float(lane % Wn) : 0.f, gy = lane < WW ? -1.0f + step * float(lane / Wn) : 0.f;
    const float ex = wave_sum(p * gx), ey = wave_sum(p * gy);
    const float vx = wave_sum(p * gx * gx) - ex * ex, vy = wave_sum(p * gy * gy) - ey * ey;
    if (lane == 0) {
        expec[3 * m + 0] = ex;
        expec[3 * m + 1] = ey;
        expec[3 * m + 2] = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
        mkpts1_f[2 * m + 0] = mkpts1_c[2 * m + 0] + ex * float(Wn / 2) * scale_px;
        mkpts1_f[2 * m + 1] = mkpts1_c[2 * m + 1] + ey * float(Wn / 2) * scale_px;
    }
}

inline int grid_for(long long total) {
    const long long b = (total + 255) / 256, cap = 64ll * pope_cu_count();
    return int(b < 1 ? 1 : (b < cap ? b : cap));
}
inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

}  // namespace

size_t pope_fine_preprocess_workspace(int M, int WW, int Cc, int Cf) {
    if (M <= 0) return 0;
    return align256(size_t(2) * M * Cc * 4) + align256(size_t(2) * M * Cf * 4) + align256(size_t(2) * M * WW * 2 * Cf * 4);
}

int pope_launch_fine_preprocess(const FinePreParams& q, hipStream_t stream) {
    if (q.M <= 0 || q.Wn <= 0 || q.Wn * q.Wn > 64 || q.stride <= 0 || (q.Cc & 31) || (q.Cf & 31) || q.Cc < 64 || q.Cf < 32) return POPE_ERR_ARG;
    const bool f32 = q.precision == POPE_PREC_F32_MFMA;
    if (!f32 && q.precision != POPE_PREC_F16X3) return POPE_ERR_ARG;
    if (!q.f0 || !q.f1 || !q.fc0 || !q.fc1 || !q.b_ids || !q.i_ids || !q.j_ids || !q.out || !q.ws) return POPE_ERR_ARG;
    if (f32 ? (!q.down_w || !q.merge_w) : (!q.down_wp || !q.merge_wp)) return POPE_ERR_ARG;
    const int WW = q.Wn * q.Wn;
    if (q.ws_bytes < pope_fine_preprocess_workspace(q.M, WW, q.Cc, q.Cf)) return POPE_ERR_WORKSPACE;
    char* ws = static_cast<char*>(q.ws);
    _Float16* cpl = reinterpret_cast<_Float16*>(ws); ws += align256(size_t(2) * q.M * q.Cc * 4);
    float* c_win = reinterpret_cast<float*>(ws); ws += align256(size_t(2) * q.M * q.Cf * 4);
    _Float16* mpl = reinterpret_cast<_Float16*>(ws);
    // POPE_PREC_F32_MFMA (the range guard's re-run): fp32 rows in the same buffers, both Linears on gemm_f32.hip
    if (f32) hipLaunchKernelGGL(fine_gather_coarse_kernel<false>, dim3(grid_for(2ll * q.M * q.Cc / 4)), dim3(256), 0, stream, q.fc0, q.fc1,
                                q.b_ids, q.i_ids, q.j_ids, q.M, q.L, q.S, q.Cc, cpl, q.range_flag);
    else hipLaunchKernelGGL(fine_gather_coarse_kernel<true>, dim3(grid_for(2ll * q.M * q.Cc / 4)), dim3(256), 0, stream, q.fc0, q.fc1,
                            q.b_ids, q.i_ids, q.j_ids, q.M, q.L, q.S, q.Cc, cpl, q.range_flag);
    int rc = pope_check_launch();
    if (rc) return rc;
    GemmParams g = {};
    g.a_pl = cpl; g.w_pl = q.down_wp; g.bias = q.down_b; g.C = c_win;
    g.A = reinterpret_cast<const float*>(cpl); g.W = q.down_w;
    g.M = 2 * q.M; g.N = q.Cf; g.K = q.Cc; g.lda = q.Cc; g.ldw = q.Cc; g.ldc = q.Cf;
    g.epilogue = EPI_BIAS; g.nbatch = 1;
    if ((rc = f32 ? pope_launch_gemm_nt_f32(g, stream) : pope_launch_gemm_nt_f16x3_planes(g, stream))) return rc;
    FineMap m0 = {q.f0, q.s0[0], q.s0[1], q.s0[2], q.s0[3], q.H0, q.W0, q.wc0};
    FineMap m1 = {q.f1, q.s1[0], q.s1[1], q.s1[2], q.s1[3], q.H1, q.W1, q.wc1};
    if (f32) hipLaunchKernelGGL(fine_gather_windows_kernel<false>, dim3(grid_for(2ll * q.M * WW * 2 * q.Cf / 4)), dim3(256), 0, stream, m0, m1,
                                c_win, q.b_ids, q.i_ids, q.j_ids, q.M, q.Wn, q.stride, q.Cf, mpl, q.range_flag);
    else hipLaunchKernelGGL(fine_gather_windows_kernel<true>, dim3(grid_for(2ll * q.M * WW * 2 * q.Cf / 4)), dim3(256), 0, stream, m0, m1,
                            c_win, q.b_ids, q.i_ids, q.j_ids, q.M, q.Wn, q.stride, q.Cf, mpl, q.range_flag);
    if ((rc = pope_check_launch())) return rc;
    GemmParams h = {};
    h.a_pl = mpl; h.w_pl = q.merge_wp; h.bias = q.merge_b; h.C = q.out;
    h.A = reinterpret_cast<const float*>(mpl); h.W = q.merge_w;
    h.M = 2 * q.M * WW; h.N = q.Cf; h.K = 2 * q.Cf; h.lda = 2 * q.Cf; h.ldw = 2 * q.Cf; h.ldc = q.Cf;
    h.epilogue = EPI_BIAS; h.nbatch = 1;
    return f32 ? pope_launch_gemm_nt_f32(h, stream) : pope_launch_gemm_nt_f16x3_planes(h, stream);
}

int pope_launch_fine_match(const float* win0, const float* win1, int M, int Wn, int C, const float* mkpts1_c, float scale_px,
                           float* expec, float* mkpts1_f, hipStream_t stream) {
    if (!win0 || !win1 || !mkpts1_c || !expec || !mkpts1_f || M <= 0 || Wn <= 0 || Wn * Wn > 64 || C <= 0) return POPE_ERR_ARG;
    hipLaunchKernelGGL(fine_match_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, win0, win1, M, Wn, C, mkpts1_c, scale_px, expec, mkpts1_f);
    return pope_check_launch();
}
