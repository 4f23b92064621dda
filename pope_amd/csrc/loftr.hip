// LoFTR encoder layer (first slice: the coarse and fine `LocalFeatureTransformer`s of the reference
// Matcher) on HIP kernels.  One layer update, src/matcher/loftr_module/transformer.py:35-58 with the linear attention
// of linear_attention.py:20-47:
//     q = x Wq^T ; k = s Wk^T ; v = s Wv^T                       (bias-free, 8 heads x D, D = C / 8)
//     Q = elu(q) + 1 ; K = elu(k) + 1 ; V = v / S                (the reference's /S ... *S is kept: it rounds)
//     KV[h] = K[h]^T V[h]  (D x D per head and image) ; Ksum[h] = sum_s K[s, h]
//     msg = (Q . KV) * 1 / (Q . Ksum + 1e-6) * S
//     msg = LayerNorm(msg Wm^T)  (eps 1e-5)
//     x  <- x + LayerNorm(relu(cat[x, msg] W0^T) W1^T)
// The five projections run on the f16x3 planes GEMM (gemm_planes.hip: fp32-equivalent arithmetic, bias-free, the MLP's
// ReLU in the epilogue, its output straight to planes); everything around them is small and HBM / launch bound:
// attention is O(L) — the per-head D x D state is reduced over the source rows in chunks (fixed order: deterministic),
// then applied row by row.  The launch sequence lives behind ONE C-ABI call per layer update (capi.hip), so a
// 4 x (self, cross) transformer is 16 calls instead of ~200 torch ops.
#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 f16x2l __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4l __attribute__((ext_vector_type(4)));

constexpr int LA_CHUNK = 64;   // source rows per workgroup of the KV reduction

// elu(x) + 1 as torch evaluates it in fp32: x > 0 ? x : exp(x) - 1, then + 1 (two roundings on the negative side)
__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expf(x) - 1.0f) + 1.0f; }

// Partial KV / Ksum of one (image, head) over one chunk of source rows.  kv: [n, S, 2C] fp32 (k | v).
// part: [n * H, chunks, D * D + D].
template <int D>
__global__ __launch_bounds__(256) void linattn_reduce_kernel(const float* __restrict__ kv, int S, int C, int H, float inv_len_unused,
                                                              float* __restrict__ part, int chunks) {
    __shared__ float Kt[LA_CHUNK][D + 1];
    __shared__ float Vt[LA_CHUNK][D];
    constexpr int VPT = D * D / 256 > 0 ? D * D / 256 : 1;   // outputs per thread: 4 (D = 32) or 1 (D = 16)
    const int c = blockIdx.y, nh = blockIdx.x, n = nh / H, h = nh - n * H;
    const int s0 = c * LA_CHUNK, ns = S - s0 < LA_CHUNK ? S - s0 : LA_CHUNK;
    const float* base = kv + (size_t(n) * S + s0) * 2 * C + h * D;
    const float len = float(S);
    for (int i = threadIdx.x; i < LA_CHUNK * D; i += 256) {
        const int s = i / D, d = i - s * D;
        float kk = 0.f, vv = 0.f;
        if (s < ns) {
            kk = elu1(base[size_t(s) * 2 * C + d]);
            vv = base[size_t(s) * 2 * C + C + d] / len;   // values / v_length (linear_attention.py:41)
        }
        Kt[s][d] = kk;
        Vt[s][d] = vv;
    }
    __syncthreads();
    float* out = part + (size_t(nh) * chunks + c) * (D * D + D);
    if (threadIdx.x < D * D / VPT) {
        const int d = threadIdx.x / (D / VPT), v0 = (threadIdx.x % (D / VPT)) * VPT;
        float acc[VPT];
#pragma unroll
        for (int e = 0; e < VPT; ++e) acc[e] = 0.f;
        float ks = 0.f;
        for (int s = 0; s < ns; ++s) {
            const float kd = Kt[s][d];
            ks += kd;
#pragma unroll
            for (int e = 0; e < VPT; ++e) acc[e] += kd * Vt[s][v0 + e];
        }
#pragma unroll
        for (int e = 0; e < VPT; ++e) out[d * D + v0 + e] = acc[e];
        if (v0 == 0) out[D * D + d] = ks;
    }
}

// KV / Ksum = sum of the chunk partials, in chunk order.  kvf: [n * H, D * D + D]
__global__ __launch_bounds__(256) void linattn_finish_kernel(const float* __restrict__ part, int chunks, int per, float* __restrict__ kvf) {
    const int nh = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;   // one value per thread: (n H) x ceil(per / 256) blocks
    if (i >= per) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(size_t(nh) * chunks + c) * per + i];
    kvf[size_t(nh) * per + i] = s;
}

// msg[l, h, v] = (sum_d Q[l,h,d] KV[h][d][v]) * (1 / (sum_d Q[l,h,d] Ksum[h][d] + eps)) * S,  Q = elu(q) + 1.
// Block: 256 threads = 256 / C rows at a time; a thread owns one output column and keeps its KV column and its head's
// Ksum in registers across the block's rows.
// PLANES (round 4): the message leaves as activation planes — the merge GEMM's operand — instead of fp32 + a split_planes pass
// (the same hi / lo arithmetic: bit-identical planes, one launch and two passes over the tensor less per layer)
template <int D, bool PLANES>
__global__ __launch_bounds__(256) void linattn_apply_kernel(const float* __restrict__ q, const float* __restrict__ kvf, int L, int C, int H,
                                                             int S, float eps, float* __restrict__ msg, int rows_per_block,
                                                             unsigned* range_flag) {
    __shared__ float Qs[2][256];
    const int rpp = 256 / C;                          // rows per pass (C = 256: 1, C = 128: 2)
    const int sub = threadIdx.x / C, col = threadIdx.x - sub * C;
    const int h = col / D, v = col - h * D;
    const int n = blockIdx.x, l0 = blockIdx.y * rows_per_block;
    const float* kvh = kvf + (size_t(n) * H + h) * (D * D + D);
    float kvc[D], ksum[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        kvc[d] = kvh[d * D + v];
        ksum[d] = kvh[D * D + d];
    }
    const float len = float(S);
    float amax = 0.f;
    for (int r = 0; r < rows_per_block; r += rpp) {
        const int l = l0 + r + sub;
        const bool ok = l < L && r + sub < rows_per_block;
        __syncthreads();
        Qs[0][threadIdx.x] = ok ? elu1(q[(size_t(n) * L + l) * C + col]) : 0.f;
        __syncthreads();
        const float* qh = &Qs[0][sub * C + h * D];
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            num += qh[d] * kvc[d];
            den += qh[d] * ksum[d];
        }
        const float m = num * (1.0f / (den + eps)) * len;
        if constexpr (PLANES) {
            if (ok) {
                const float sv = m * K_PLANES_ACT_SCALE;
                amax = fmaxf(amax, fabsf(sv));
                if (!(sv == sv)) amax = INFINITY;
                const _Float16 hi = _Float16(sv), lo = _Float16(sv - float(hi));
                _Float16* o = reinterpret_cast<_Float16*>(msg) + (size_t(n) * L + l) * 2 * C + (col >> 5) * 64 + (col & 31);
                o[0] = hi;
                o[32] = lo;
            }
        } else {
            if (ok) msg[(size_t(n) * L + l) * C + col] = m;
        }
    }
    if constexpr (PLANES) pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax < POPE_F16_OVERFLOW));
}

// planes row [x | LayerNorm(m) * w + b] of width 2C (the MLP's input cat[x, message], transformer.py:54), one wave per
// row, NV = C / 64 consecutive columns per lane
template <int NV, bool PLANES>   // PLANES = false: the fp32 twin writes [x | LN(m)] as floats [rows, 2C]
__global__ __launch_bounds__(256) void ln_cat_planes_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             _Float16* __restrict__ pl, int rows, float eps, float scale, unsigned* flag) {
    constexpr int C = NV * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c0 = lane * NV;
    float mv[NV], xv[NV];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        mv[e] = m[size_t(row) * C + c0 + e];
        xv[e] = x[size_t(row) * C + c0 + e];
        s += mv[e];
    }
    const float mean = wave_sum(s) * (1.0f / float(C));
    float qv = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float d = mv[e] - mean;
        qv += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(qv) * (1.0f / float(C)) + eps);
    _Float16* pr = pl + size_t(row) * 4 * C;   // 2 * (2C) halves per row
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int c = c0 + e;
        const float vals[2] = {xv[e] * scale, ((mv[e] - mean) * rstd * w[c] + b[c]) * scale};
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int cc = part * C + c;
            const float val = vals[part];
            if constexpr (PLANES) {
                amax = fmaxf(amax, fabsf(val));
                const _Float16 hi = _Float16(val);
                pr[(cc >> 5) * 64 + (cc & 31)] = hi;
                pr[(cc >> 5) * 64 + 32 + (cc & 31)] = _Float16(val - float(hi));
            } else {
                reinterpret_cast<float*>(pl)[size_t(row) * 2 * C + cc] = val;   // scale = 1
            }
        }
    }
    if constexpr (PLANES) pope_range_flag(flag, POPE_RANGE_LAYERNORM, !(amax < POPE_F16_OVERFLOW) || !(fabsf(mean) + rstd < INFINITY));
}

// x <- x + LayerNorm(y) * w + b   (transformer.py:56-58), one wave per row
template <int NV>
__global__ __launch_bounds__(256) void ln_add_kernel(float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w,
                                                      const float* __restrict__ b, int rows, float eps) {
    constexpr int C = NV * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c0 = lane * NV;
    float yv[NV];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        yv[e] = y[size_t(row) * C + c0 + e];
        s += yv[e];
    }
    const float mean = wave_sum(s) * (1.0f / float(C));
    float qv = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float d = yv[e] - mean;
        qv += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(qv) * (1.0f / float(C)) + eps);
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int c = c0 + e;
        x[size_t(row) * C + c] += (yv[e] - mean) * rstd * w[c] + b[c];
    }
}

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

size_t pope_loftr_layer_workspace(int n, int L, int S, int C, int H) {
    const size_t rx = size_t(n) * L, rs = size_t(n) * S, D = C / H, chunks = (S + LA_CHUNK - 1) / LA_CHUNK;
    return al(rx * C * 4) /*xp*/ + al(rs * C * 4) /*sp*/ + al(rx * C * 4) /*q, later merge out*/ + al(rs * 2 * C * 4) /*kv*/ +
           al(size_t(n) * H * chunks * (D * D + D) * 4) + al(size_t(n) * H * (D * D + D) * 4) + al(rx * C * 4) /*msg, later mlp out*/ +
           al(rx * C * 4) /*msg planes*/ + al(rx * 2 * C * 4) /*cat planes*/ + al(rx * 2 * C * 4) /*hidden planes*/;
}

int pope_launch_loftr_layer(const LoftrLayerParams& p, hipStream_t stream) {
    const int C = p.C, H = p.H, D = C / H;
    if (!p.x || !p.source || !p.ws || p.n <= 0 || p.L <= 0 || p.S <= 0 || (C != 256 && C != 128) || H != 8) return POPE_ERR_ARG;
    const bool f32 = p.precision == POPE_PREC_F32_MFMA;
    if (!f32 && p.precision != POPE_PREC_F16X3) return POPE_ERR_ARG;
    if (!p.norm1_w || !p.norm1_b || !p.norm2_w || !p.norm2_b) return POPE_ERR_ARG;
    if (f32 ? (!p.q_w || !p.kv_w || !p.merge_w || !p.mlp0_w || !p.mlp1_w) : (!p.q_wp || !p.kv_wp || !p.merge_wp || !p.mlp0_wp || !p.mlp1_wp))
        return POPE_ERR_ARG;
    if (p.ws_bytes < pope_loftr_layer_workspace(p.n, p.L, p.S, C, H)) return POPE_ERR_WORKSPACE;
    const size_t rx = size_t(p.n) * p.L, rs = size_t(p.n) * p.S;
    if (rx > 0x7fffffffull / (2 * C) || rs > 0x7fffffffull / (2 * C)) return POPE_ERR_ARG;
    const int chunks = (p.S + LA_CHUNK - 1) / LA_CHUNK, per = D * D + D;
    char* w = static_cast<char*>(p.ws);
    auto take = [&](size_t bytes) { char* r = w; w += al(bytes); return r; };
    void* xp = take(rx * C * 4);
    void* sp = take(rs * C * 4);
    float* q = reinterpret_cast<float*>(take(rx * C * 4));
    float* kv = reinterpret_cast<float*>(take(rs * 2 * C * 4));
    float* part = reinterpret_cast<float*>(take(size_t(p.n) * H * chunks * per * 4));
    float* kvf = reinterpret_cast<float*>(take(size_t(p.n) * H * per * 4));
    float* msg = reinterpret_cast<float*>(take(rx * C * 4));
    void* msgp = take(rx * C * 4);
    void* catp = take(rx * 2 * C * 4);
    void* hidp = take(rx * 2 * C * 4);
    const bool self = p.source == p.x && p.S == p.L;
    int rc;
#define LT(call) do { if ((rc = (call))) return rc; } while (0)
    // POPE_PREC_F32_MFMA (the range guard's re-run): the same sequence with fp32 operands on gemm_f32.hip — "planes" buffers
    // hold plain fp32 rows (same bytes), no operand split, no range contract
    auto gemm = [&](const void* a_pl, const void* w_pl, float* Cf, void* Cp, int M, int N, int K, int epi) {
        GemmParams g = {};
        if (f32) {
            g.A = static_cast<const float*>(a_pl); g.W = static_cast<const float*>(w_pl);
            g.C = Cf ? Cf : static_cast<float*>(Cp);
            g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
            g.epilogue = EPI_CONV; g.act_slope = epi == EPI_BIAS_RELU ? 0.f : 1.f;
            return pope_launch_gemm_nt_f32(g, stream);
        }
        g.a_pl = a_pl; g.w_pl = w_pl; g.C = Cf; g.c_pl = Cp;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
        g.epilogue = epi;
        g.range_flag = p.range_flag; g.range_bit = POPE_RANGE_GELU;
        return pope_launch_gemm_nt_f16x3_planes(g, stream);
    };
    // 1. operands of the projections
    if (f32) {
        xp = p.x;
        sp = const_cast<float*>(p.source);
    } else {
        LT(pope_launch_split_planes(p.x, xp, int(rx), C, K_PLANES_ACT_SCALE, p.range_flag, stream));
        if (!self) LT(pope_launch_split_planes(p.source, sp, int(rs), C, K_PLANES_ACT_SCALE, p.range_flag, stream));
    }
    const void* Wq = f32 ? static_cast<const void*>(p.q_w) : p.q_wp;
    const void* Wkv = f32 ? static_cast<const void*>(p.kv_w) : p.kv_wp;
    const void* Wm = f32 ? static_cast<const void*>(p.merge_w) : p.merge_wp;
    const void* W0 = f32 ? static_cast<const void*>(p.mlp0_w) : p.mlp0_wp;
    const void* W1 = f32 ? static_cast<const void*>(p.mlp1_w) : p.mlp1_wp;
    // 2. q = x Wq^T ; [k | v] = source [Wk ; Wv]^T
    LT(gemm(xp, Wq, q, nullptr, int(rx), C, C, EPI_BIAS));
    LT(gemm(self ? xp : sp, Wkv, kv, nullptr, int(rs), 2 * C, C, EPI_BIAS));
    // 3. per-head state, 4. message
    if (D == 32) hipLaunchKernelGGL(linattn_reduce_kernel<32>, dim3(p.n * H, chunks), dim3(256), 0, stream, kv, p.S, C, H, 0.f, part, chunks);
    else hipLaunchKernelGGL(linattn_reduce_kernel<16>, dim3(p.n * H, chunks), dim3(256), 0, stream, kv, p.S, C, H, 0.f, part, chunks);
    hipLaunchKernelGGL(linattn_finish_kernel, dim3(p.n * H, (per + 255) / 256), dim3(256), 0, stream, part, chunks, per, kvf);
    const int rpb = 16;
    // (f16x3: the message is written as planes straight away; fp32 mode: plain rows)
#define LA_APPLY(DD, PL, DST) hipLaunchKernelGGL((linattn_apply_kernel<DD, PL>), dim3(p.n, (p.L + rpb - 1) / rpb), dim3(256), 0, stream, q, kvf, p.L, \
                                                 C, H, p.S, 1e-6f, DST, rpb, p.range_flag)
    if (f32) {
        if (D == 32) LA_APPLY(32, false, msg); else LA_APPLY(16, false, msg);
        msgp = msg;
    } else {
        if (D == 32) LA_APPLY(32, true, static_cast<float*>(msgp)); else LA_APPLY(16, true, static_cast<float*>(msgp));
    }
#undef LA_APPLY
    // 5. merge (-> q buffer), 6. cat[x, LN1(merge)] as planes
    LT(gemm(msgp, Wm, q, nullptr, int(rx), C, C, EPI_BIAS));
    const dim3 rows4(unsigned((rx + 3) / 4));
    const float cat_scale = f32 ? 1.0f : K_PLANES_ACT_SCALE;
#define LN_CAT(NV, PL) hipLaunchKernelGGL((ln_cat_planes_kernel<NV, PL>), rows4, dim3(256), 0, stream, p.x, q, p.norm1_w, p.norm1_b, \
                                          static_cast<_Float16*>(catp), int(rx), p.ln_eps, cat_scale, p.range_flag)
    if (C == 256) { if (f32) LN_CAT(4, false); else LN_CAT(4, true); }
    else { if (f32) LN_CAT(2, false); else LN_CAT(2, true); }
#undef LN_CAT
    // 7. MLP: relu(cat W0^T) -> planes ; W1 -> fp32 (msg buffer) ; 8. x += LN2(.)
    LT(gemm(catp, W0, nullptr, hidp, int(rx), 2 * C, 2 * C, EPI_BIAS_RELU));
    LT(gemm(hidp, W1, msg, nullptr, int(rx), C, 2 * C, EPI_BIAS));
    if (C == 256) hipLaunchKernelGGL(ln_add_kernel<4>, rows4, dim3(256), 0, stream, p.x, msg, p.norm2_w, p.norm2_b, int(rx), p.ln_eps);
    else hipLaunchKernelGGL(ln_add_kernel<2>, rows4, dim3(256), 0, stream, p.x, msg, p.norm2_w, p.norm2_b, int(rx), p.ln_eps);
#undef LT
    return pope_check_launch();
}
