// SAM image encoder, POPE_PREC_F32_MFMA: the range guard's re-run and the strict-fp32 mode of ImageEncoderViT
// (segment_anything/segment_anything/modeling/image_encoder.py:107-118).  Same launch sequence as sam.hip with fp32
// operands everywhere: every Linear / convolution on gemm_f32.hip (v_mfma_f32_32x32x2_f32: the reference's own fp32
// arithmetic, no range contract), LayerNorm by layernorm.hip's fp32 kernel, and a plain fp32 attention kernel for the
// window / global blocks with the decomposed relative-position terms (image_encoder.py:217-235, 325-358).  It shares the
// f16x3 path's workspace (the operand-plane regions serve as scratch).  Speed is not a goal here — this is the path a
// range-guard event falls back to, ~10x slower than f16x3 — correctness against the reference fixtures is
// (tests/test_gpu_sam.py).
#include "common.h"
#include "kernels.h"

namespace {

inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }
inline int grid_for(long long total, int per_block = 256) {
    const long long b = (total + per_block - 1) / per_block, cap = 64ll * pope_cu_count();
    return int(b < 1 ? 1 : (b < cap ? b : cap));
}

// image [B, 3, S, S] -> rows [B g g, 3 P P] in the order of Conv2d's weight.reshape(dim, -1) (c, ky, kx); P % 4 == 0
__global__ __launch_bounds__(256) void sam32_im2col_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int S, int P) {
    const int g = S / P, K = 3 * P * P, pieces = K / 4;
    const long long total = (long long)B * g * g * pieces;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int pc = int(id % pieces);
        const long long row = id / pieces;
        const int px = int(row % g), py = int((row / g) % g), b = int(row / ((long long)g * g));
        const int k = pc * 4, c = k / (P * P), ky = (k - c * P * P) / P, kx = k - c * P * P - ky * P;
        *reinterpret_cast<f32x4*>(out + (size_t)row * K + k) =
            *reinterpret_cast<const f32x4*>(img + (((size_t)b * 3 + c) * S + (py * P + ky)) * S + px * P + kx);
    }
}

// Attention of one block in fp32 (image_encoder.py:217-235): qkv [B g g, 3 dim] (column = which * dim + head * HD + c), windows
// of ws x ws tokens (ws = g: one global window), zero-padded at the bottom / right: a padded token's k and v are the qkv
// bias (Linear of the zero row norm1's padding leaves, image_encoder.py:169-176), padded queries are dropped.
//   score(q, k) = hd^-1/2 q.k + q.Rh[qy][ky] + q.Rw[qx][kx]      (the unscaled q in the position terms, :225-231, 325-358)
// One thread per query (64 per workgroup), keys in tiles of 32 through LDS (broadcast reads), online softmax.
constexpr int ATT_KT = 32;
template <int HD>
__global__ __launch_bounds__(64) void sam32_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                        const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                        float* __restrict__ out, int B, int g, int ws, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    const int dim = heads * HD, nw = (g + ws - 1) / ws, Nw = ws * ws;
    const int qtiles = (Nw + 63) / 64;
    const int qt = blockIdx.x % qtiles;
    int rest = blockIdx.x / qtiles;
    const int head = rest % heads; rest /= heads;
    const int wx = rest % nw; rest /= nw;
    const int wy = rest % nw, b = rest / nw;
    float* Ks = smem32;                         // [ATT_KT][HD]
    float* Vs = Ks + ATT_KT * HD;               // [ATT_KT][HD]
    float* Rq = Vs + ATT_KT * HD;               // [64][2 ws + 1]: q.Rh[qy][.] then q.Rw[qx][.] of this thread's query
    const int rq_pitch = 2 * ws + 1;
    const int tid = threadIdx.x;
    const int qi = qt * 64 + tid;               // query index inside the window
    const int qy = qi / ws, qx = qi - qy * ws;
    const int ty = wy * ws + qy, tx = wx * ws + qx;
    const bool real_q = qi < Nw && ty < g && tx < g;
    float q[HD], o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) { q[c] = 0.f; o[c] = 0.f; }
    if (real_q) {
        const float* src = qkv + ((size_t)b * g * g + (size_t)ty * g + tx) * 3 * dim + head * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
            q[c] = v[0]; q[c + 1] = v[1]; q[c + 2] = v[2]; q[c + 3] = v[3];
        }
    }
    // position terms of this query against every key row / key column of the window
    float* myR = Rq + tid * rq_pitch;
    for (int k = 0; k < ws; ++k) {
        float ah = 0.f, aw = 0.f;
        if (real_q && rel_h) {
            const float* rh = rel_h + ((size_t)qy * ws + k) * HD;
            const float* rw = rel_w + ((size_t)qx * ws + k) * HD;
#pragma unroll
            for (int c = 0; c < HD; ++c) { ah += q[c] * rh[c]; aw += q[c] * rw[c]; }
        }
        myR[k] = ah;
        myR[ws + k] = aw;
    }
    const float scale = 1.0f / sqrtf(float(HD));
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < Nw; k0 += ATT_KT) {
        __syncthreads();
        for (int idx = tid; idx < ATT_KT * (HD / 4); idx += 64) {
            const int j = idx / (HD / 4), c = (idx - j * (HD / 4)) * 4;
            const int kk = k0 + j;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (kk < Nw) {
                const int ky = kk / ws, kx = kk - ky * ws, yy = wy * ws + ky, xx = wx * ws + kx;
                if (yy < g && xx < g) {
                    const float* src = qkv + ((size_t)b * g * g + (size_t)yy * g + xx) * 3 * dim + head * HD + c;
                    kv = *reinterpret_cast<const f32x4*>(src + dim);
                    vv = *reinterpret_cast<const f32x4*>(src + 2 * dim);
                } else {   // a padded token: Linear(0) = bias
                    kv = *reinterpret_cast<const f32x4*>(qkv_bias + dim + head * HD + c);
                    vv = *reinterpret_cast<const f32x4*>(qkv_bias + 2 * dim + head * HD + c);
                }
            }
            *reinterpret_cast<f32x4*>(Ks + j * HD + c) = kv;
            *reinterpret_cast<f32x4*>(Vs + j * HD + c) = vv;
        }
        __syncthreads();
        const int nk = Nw - k0 < ATT_KT ? Nw - k0 : ATT_KT;
        for (int j = 0; j < nk; ++j) {
            const int kk = k0 + j, ky = kk / ws, kx = kk - ky * ws;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) s += q[c] * Ks[j * HD + c];
            s = s * scale + myR[ky] + myR[ws + kx];
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), p = expf(s - mn);   // m = -inf on the first key: alpha = 0, l and o are 0
            l = l * alpha + p;
#pragma unroll
            for (int c = 0; c < HD; ++c) o[c] = o[c] * alpha + p * Vs[j * HD + c];
            m = mn;
        }
    }
    if (real_q) {
        const float inv = 1.0f / l;
        float* dst = out + ((size_t)b * g * g + (size_t)ty * g + tx) * dim + head * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(dst + c) = f32x4{o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
    }
}

// LayerNorm2d (common.py:27-43: per pixel over the channels, eps inside the sqrt, a true division), one wave per pixel.
// BORDERED_OUT: in [B g g, C] -> out fp32 [B, g + 2, g + 2, C] with a zero border (the 3x3 convolution's operand);
// else: in fp32 [B, g + 2, g + 2, C] (the convolution's bordered output) -> out NCHW [B, C, g, g].
template <bool BORDERED_OUT>
__global__ __launch_bounds__(256) void sam32_ln2d_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bvec,
                                                         float* __restrict__ out, int B, int g, int C, float eps) {
    const int gp = g + 2, lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, total = (long long)B * (BORDERED_OUT ? gp * gp : g * g);
    for (long long pix = blockIdx.x * 4ll + (threadIdx.x >> 6); pix < total; pix += waves) {
        int b, y, x;
        if (BORDERED_OUT) {
            b = int(pix / (gp * gp));
            const int rem = int(pix - (long long)b * gp * gp);
            y = rem / gp - 1; x = rem % gp - 1;
        } else {
            b = int(pix / (g * g));
            const int rem = int(pix - (long long)b * g * g);
            y = rem / g; x = rem % g;
        }
        const bool interior = y >= 0 && y < g && x >= 0 && x < g;
        const float* src = BORDERED_OUT ? in + ((size_t)b * g * g + (size_t)y * g + x) * C
                                        : in + ((size_t)b * gp * gp + (size_t)(y + 1) * gp + (x + 1)) * C;
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += interior ? src[c] : 0.f;
        const float u = wave_sum(sum) / float(C);
        float sq = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = interior ? src[c] - u : 0.f;
            sq += d * d;
        }
        const float den = sqrtf(wave_sum(sq) / float(C) + eps);
        for (int c = lane; c < C; c += 64) {
            const float r = interior ? w[c] * ((src[c] - u) / den) + bvec[c] : 0.f;
            if (BORDERED_OUT) out[(size_t)pix * C + c] = r;
            else out[(((size_t)b * C + c) * g + y) * g + x] = r;
        }
    }
}

}  // namespace

int pope_launch_sam_encoder_f32mfma(const SamEncParams& q, hipStream_t stream) {
    const int g = q.img / q.patch, hd = q.dim / q.heads, dim = q.dim, hidden = q.hidden, oc = q.out_chans;
    const int kp = 3 * q.patch * q.patch;
    if ((q.patch & 3) || (hd != 64 && hd != 80)) return POPE_ERR_ARG;
    const int rows = q.B * g * g;
    const size_t gp = size_t(g) + 2, brows = size_t(q.B) * gp * gp;
    // the workspace of the f16x3 path (pope_sam_encoder_workspace), re-read as: x | xn | big | scratch (the operand sets) ...
    char* base = static_cast<char*>(q.ws);
    char* const ws_end = base + q.ws_bytes;
    auto take = [&](size_t bytes) { char* p = base; base += align256(bytes); return p; };
    float* x = reinterpret_cast<float*>(take(size_t(rows) * dim * 4));
    float* xn = reinterpret_cast<float*>(take(size_t(rows) * dim * 4));
    size_t big_bytes = size_t(rows) * 4 * dim * 4;
    if (size_t(rows) * hidden * 4 > big_bytes) big_bytes = size_t(rows) * hidden * 4;
    if (size_t(rows) * kp * 4 > big_bytes) big_bytes = size_t(rows) * kp * 4;
    float* big = reinterpret_cast<float*>(take(big_bytes));
    // ... and, from the END (where the f16x3 path keeps them too), the neck buffers; the attention output sits between
    float* t2 = reinterpret_cast<float*>(ws_end - align256(brows * oc * 4));
    float* t1b = reinterpret_cast<float*>(reinterpret_cast<char*>(t2) - align256(brows * oc * 4));
    float* t1 = reinterpret_cast<float*>(reinterpret_cast<char*>(t1b) - align256(size_t(rows) * oc * 4));
    float* att = reinterpret_cast<float*>(base);
    if (reinterpret_cast<char*>(att) + size_t(rows) * dim * 4 > reinterpret_cast<char*>(t1)) return POPE_ERR_WORKSPACE;
    if (size_t(rows + 256) * (hidden > 3 * dim ? hidden : 3 * dim) * 4 >= (1ull << 32) - 512) return POPE_ERR_ARG;

    const float eps = q.block_eps > 0.f ? q.block_eps : 1e-6f, neck_eps = q.neck_eps > 0.f ? q.neck_eps : 1e-6f;
    int rc;
#define POPE_TRY(call) do { if ((rc = (call))) return rc; } while (0)
    auto gemm = [&](const float* A, int M, const void* W, const float* bias, float* Cf, int N, int K, int epi, const float* gamma,
                    const float* res) {
        GemmParams p = {};
        p.A = A; p.W = static_cast<const float*>(W); p.bias = bias; p.C = Cf;
        p.lda = K; p.ldw = K; p.ldc = N; p.M = M; p.N = N; p.K = K;
        p.epilogue = epi; p.gamma = gamma; p.res = res; p.ldres = N;
        return pope_launch_gemm_nt_f32(p, stream);
    };
    // patch embed + absolute position table (image_encoder.py:108-110)
    hipLaunchKernelGGL(sam32_im2col_kernel, dim3(grid_for((long long)rows * (kp / 4))), dim3(256), 0, stream, q.image, big, q.B, q.img, q.patch);
    POPE_TRY(pope_check_launch());
    for (int b = 0; b < q.B; ++b) {   // the position table is per token, the same for every image: one GEMM per image
        const size_t r0 = size_t(b) * g * g;
        if (q.pos) POPE_TRY(gemm(big + r0 * kp, g * g, q.patch_wp, q.patch_b, x + r0 * dim, dim, kp, EPI_BIAS_LS_RES, q.ones, q.pos));
        else POPE_TRY(gemm(big + r0 * kp, g * g, q.patch_wp, q.patch_b, x + r0 * dim, dim, kp, EPI_BIAS, nullptr, nullptr));
    }
    for (int i = 0; i < q.depth; ++i) {
        const SamBlockParams& k = q.blocks[i];
        const int ws = (k.global || q.window <= 0) ? g : q.window;
        // x = x + attn(norm1(x))                                         image_encoder.py:166-179
        POPE_TRY(pope_launch_layernorm_f32(x, dim, k.norm1_w, k.norm1_b, xn, dim, rows, dim, eps, stream));
        POPE_TRY(gemm(xn, rows, k.qkv_wp, k.qkv_b, big, 3 * dim, dim, EPI_BIAS, nullptr, nullptr));
        {
            const int nw = (g + ws - 1) / ws, qtiles = (ws * ws + 63) / 64;
            const long long blocks = (long long)q.B * nw * nw * q.heads * qtiles;
            if (blocks > 0x7fffffffll) return POPE_ERR_ARG;
            const size_t lds = size_t(2 * ATT_KT * hd + 64 * (2 * ws + 1)) * sizeof(float);
            if (lds > 64 * 1024) return POPE_ERR_ARG;
            if (hd == 80) hipLaunchKernelGGL(sam32_attn_kernel<80>, dim3((unsigned)blocks), dim3(64), lds, stream, big, k.qkv_b, k.rel_h, k.rel_w, att, q.B, g, ws, q.heads);
            else hipLaunchKernelGGL(sam32_attn_kernel<64>, dim3((unsigned)blocks), dim3(64), lds, stream, big, k.qkv_b, k.rel_h, k.rel_w, att, q.B, g, ws, q.heads);
            POPE_TRY(pope_check_launch());
        }
        POPE_TRY(gemm(att, rows, k.proj_wp, k.proj_b, x, dim, dim, EPI_BIAS_LS_RES, q.ones, x));
        // x = x + mlp(norm2(x))                                          image_encoder.py:181; common.py:13-25
        POPE_TRY(pope_launch_layernorm_f32(x, dim, k.norm2_w, k.norm2_b, xn, dim, rows, dim, eps, stream));
        POPE_TRY(gemm(xn, rows, k.fc1_wp, k.fc1_b, big, hidden, dim, EPI_BIAS_GELU, nullptr, nullptr));
        POPE_TRY(gemm(big, rows, k.fc2_wp, k.fc2_b, x, dim, hidden, EPI_BIAS_LS_RES, q.ones, x));
        for (int t = 0; t < q.n_taps; ++t)
            if (q.tap_blocks[t] == i && q.tap_out[t] &&
                hipMemcpyAsync(q.tap_out[t], x, size_t(rows) * dim * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return POPE_ERR_LAUNCH;
    }
    // neck (image_encoder.py:89-105): 1x1 conv (no bias) -> LayerNorm2d -> 3x3 conv pad 1 (no bias) -> LayerNorm2d
    POPE_TRY(gemm(x, rows, q.neck0_wp, nullptr, t1, oc, dim, EPI_BIAS, nullptr, nullptr));
    hipLaunchKernelGGL(sam32_ln2d_kernel<true>, dim3(grid_for((long long)brows, 4)), dim3(256), 0, stream, t1, q.neck1_w, q.neck1_b, t1b, q.B, g,
                       oc, neck_eps);
    POPE_TRY(pope_check_launch());
    {
        GemmParams c = {};
        const int Wp = g + 2;
        const size_t shift = size_t(Wp) + 1;   // output row R is pixel R + Wp + 1 (conv.hip)
        c.A = t1b; c.W = static_cast<const float*>(q.neck2_wp); c.bias = nullptr;
        c.lda = oc; c.ldw = 9 * oc; c.ldc = oc;
        c.M = int(brows - (2 * size_t(Wp) + 2)); c.N = oc; c.K = 9 * oc;
        c.epilogue = EPI_CONV; c.act_slope = 1.0f;   // identity
        c.C = t2 + shift * oc;
        c.conv_wp = Wp;
        POPE_TRY(pope_launch_gemm_nt_f32(c, stream));
    }
    hipLaunchKernelGGL(sam32_ln2d_kernel<false>, dim3(grid_for((long long)rows, 4)), dim3(256), 0, stream, t2, q.neck3_w, q.neck3_b, q.out, q.B, g, oc,
                       neck_eps);
    POPE_TRY(pope_check_launch());
#undef POPE_TRY
    return POPE_OK;
}
