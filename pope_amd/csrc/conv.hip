// LoFTR local-feature CNN (ResNetFPN_8_2, src/matcher/backbone/resnet_fpn.py:43-118) on the f16x3 planes GEMM.
//
// Activations live as activation planes (pope_hip.h layout, scale 8) in NHWC with a ONE-PIXEL ZERO BORDER:
//     T[n][Hp = H + 2][Wp = W + 2][Cp = channels rounded up to 32],   one pixel = one planes row of Cp * 4 bytes.
// A 3x3 stride-1 convolution is then ONE planes GEMM with no im2col at all (gemm_planes.hip, CONV): output row R
// is pixel R + Wp + 1, and the A row of tap (dy, dx) is row R + dy * Wp + dx — the same shift for every row, i.e. a
// scalar offset per K-step of the loader.  Rows that are border pixels come out as garbage and are zeroed afterwards
// (`zero_border_kernel`, (2 Hp + 2 Wp) / (Hp Wp) of the rows); eval-mode BatchNorm is folded into filter and bias by the
// host (pope_amd/loftr.py), ReLU / LeakyReLU / the BasicBlock shortcut are the GEMM's epilogue (EPI_CONV).
// The three stride-2 layers (7x7 stem on the gray image, the first 3x3 and the 1x1 shortcut of layer2 / layer3) gather
// their taps into planes rows first (their outputs are 4x smaller than their inputs), 1x1 convolutions are plain GEMMs
// over the pixel rows, and the FPN's bilinear x2 (align_corners) + lateral add is one kernel that writes planes.
#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr float A_SCALE = K_PLANES_ACT_SCALE;

// 7x7 stride-2 pad-3 taps of the gray image as planes rows [n * Hp * Wp, 64] (49 taps, zero-filled to two 32-column
// chunks), one row per pixel of the zero-bordered H/2 x W/2 output grid (border rows all zero).
template <bool PLANES>   // false: the fp32 twin (rows of 64 floats, same pitch in bytes)
__global__ __launch_bounds__(256) void stem_gather_kernel(const float* __restrict__ img, _Float16* __restrict__ out,
                                                          int n, int H, int W, unsigned* range_flag) {
    const int Hp = H / 2 + 2, Wp = W / 2 + 2;
    const long long total = (long long)n * Hp * Wp * 8;   // eight 8-column pieces per row
    float amax = 0.f;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int t = int(id & 7);
        const long long row = id >> 3;
        const int xo = int(row % Wp), yo = int((row / Wp) % Hp), b = int(row / ((long long)Wp * Hp));
        const bool interior = xo >= 1 && xo < Wp - 1 && yo >= 1 && yo < Hp - 1;
        f16x8 hi, lo;
        float raw[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * t + e, ky = c / 7, kx = c - 7 * ky;
            const int y = 2 * (yo - 1) + ky - 3, x = 2 * (xo - 1) + kx - 3;
            float v = 0.f;
            if (interior && c < 49 && y >= 0 && y < H && x >= 0 && x < W) v = img[((size_t)b * H + y) * W + x];
            amax = fmaxf(amax, fabsf(v));
            if (!(v == v)) amax = INFINITY;
            raw[e] = v;
            const float s = v * A_SCALE;
            hi[e] = _Float16(s);
            lo[e] = _Float16(s - float(hi[e]));
        }
        if constexpr (PLANES) {
            _Float16* o = out + row * 128 + (t >> 2) * 64 + (t & 3) * 8;
            *reinterpret_cast<f16x8*>(o) = hi;
            *reinterpret_cast<f16x8*>(o + 32) = lo;
        } else {
            float* o = reinterpret_cast<float*>(out) + row * 64 + 8 * t;
            *reinterpret_cast<f32x4*>(o) = f32x4{raw[0], raw[1], raw[2], raw[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{raw[4], raw[5], raw[6], raw[7]};
        }
    }
    if constexpr (PLANES) pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax * A_SCALE < POPE_F16_OVERFLOW));
}

// Stride-2 taps of a zero-bordered planes tensor [n, Hpi, Wpi, cch chunks] as planes rows [n * Hpo * Wpo, taps * cch
// chunks] over the zero-bordered output grid (Hpo = (Hpi - 2) / 2 + 2): taps = 9 (3x3, pad 1; tap-major as the
// weights) or 1 (the 1x1 shortcut: the centre).  A pure 16-byte copy: planes stay planes.
__global__ __launch_bounds__(256) void gather_s2_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int n, int Hpi,
                                                        int Wpi, int cch, int taps) {
    const int Hpo = (Hpi - 2) / 2 + 2, Wpo = (Wpi - 2) / 2 + 2;
    const int per_row = taps * cch * 8;
    const long long total = (long long)n * Hpo * Wpo * per_row;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int piece = int(id % per_row);
        const long long row = id / per_row;
        const int xo = int(row % Wpo), yo = int((row / Wpo) % Hpo), b = int(row / ((long long)Wpo * Hpo));
        const int pc = piece & 7, chunk = (piece >> 3) % cch, tap = (piece >> 3) / cch;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (xo >= 1 && xo < Wpo - 1 && yo >= 1 && yo < Hpo - 1) {
            const int ky = taps == 9 ? tap / 3 : 1, kx = taps == 9 ? tap - 3 * (tap / 3) : 1;
            const int yi = 2 * (yo - 1) + ky, xi = 2 * (xo - 1) + kx;   // padded input coordinates
            v = in[(((size_t)b * Hpi + yi) * Wpi + xi) * (cch * 8) + chunk * 8 + pc];
        }
        out[id] = v;
    }
}

// zero the border pixels' rows of a [n, Hp, Wp] pixel-row tensor (row_bytes per pixel: planes or fp32, same pitch)
__global__ __launch_bounds__(256) void zero_border_kernel(u32x4* __restrict__ buf, int n, int Hp, int Wp, int row_pieces) {
    const int per_img = 2 * Wp + 2 * (Hp - 2);
    const long long total = (long long)n * per_img * row_pieces;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int piece = int(id % row_pieces);
        const long long bp = id / row_pieces;
        const int k = int(bp % per_img), b = int(bp / per_img);
        int y, x;
        if (k < Wp) { y = 0; x = k; }
        else if (k < 2 * Wp) { y = Hp - 1; x = k - Wp; }
        else { const int j = k - 2 * Wp; y = 1 + (j >> 1); x = (j & 1) ? Wp - 1 : 0; }
        buf[(((size_t)b * Hp + y) * Wp + x) * row_pieces + piece] = u32x4{0u, 0u, 0u, 0u};
    }
}

// FPN merge (resnet_fpn.py:109-115): out = lateral + bilinear_x2(src, align_corners=True), written as planes into the
// interior of a zero-bordered tensor.  lateral [n, Hp, Wp, ldl] fp32, src [n, Hsp, Wsp, lds] fp32 (half resolution,
// both zero-bordered), out planes [n, Hp, Wp, Cp]; channels c < C in groups of four (C % 4 == 0).
template <bool PLANES>   // false: fp32 rows [.., Cp]
__global__ __launch_bounds__(256) void upsample_add_planes_kernel(const float* __restrict__ lat, int ldl, const float* __restrict__ src,
                                                                  int lds, _Float16* __restrict__ out, int Cp, int C, int n, int Hp,
                                                                  int Wp, unsigned* range_flag) {
    const int H = Hp - 2, W = Wp - 2, Hs = H / 2, Ws = W / 2, Hsp = Hs + 2, Wsp = Ws + 2;
    const int groups = Cp / 4;   // the channel padding (C <= c < Cp) is written as zeros
    // torch's area_pixel_compute_scale for align_corners: (in - 1) / (out - 1), 0 for a single output pixel
    const float sh = H > 1 ? float(Hs - 1) / float(H - 1) : 0.f, sw = W > 1 ? float(Ws - 1) / float(W - 1) : 0.f;
    const long long total = (long long)n * H * W * groups;
    float amax = 0.f;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int gidx = int(id % groups);
        const long long pix = id / groups;
        const int x = int(pix % W), y = int((pix / W) % H), b = int(pix / ((long long)W * H));
        const float ry = sh * float(y), rx = sw * float(x);
        const int y0 = int(ry), x0 = int(rx);
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float ly1 = ry - float(y0), lx1 = rx - float(x0), ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const int c = 4 * gidx;
        auto at = [&](int yy, int xx) {
            return *reinterpret_cast<const f32x4*>(src + (((size_t)b * Hsp + yy + 1) * Wsp + xx + 1) * lds + c);
        };
        const size_t prow = ((size_t)b * Hp + y + 1) * Wp + x + 1;
        _Float16* o = out + prow * 2 * Cp + (c >> 5) * 64 + (c & 31);
        float* of = reinterpret_cast<float*>(out) + prow * Cp + c;
        if (c >= C) {
            if constexpr (PLANES) {
                *reinterpret_cast<f16x4*>(o) = f16x4{0, 0, 0, 0};
                *reinterpret_cast<f16x4*>(o + 32) = f16x4{0, 0, 0, 0};
            } else {
                *reinterpret_cast<f32x4*>(of) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            continue;
        }
        const f32x4 v00 = at(y0, x0), v01 = at(y0, x1), v10 = at(y1, x0), v11 = at(y1, x1);
        const f32x4 l = *reinterpret_cast<const f32x4*>(lat + prow * ldl + c);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // upsample_bilinear2d: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
            const float up = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
            v[e] = l[e] + up;
            amax = fmaxf(amax, fabsf(v[e]));
            if (!(v[e] == v[e])) amax = INFINITY;
        }
        if constexpr (PLANES) {
            f16x4 hi, lo;
            pope_split4(v * A_SCALE, hi, lo);
            *reinterpret_cast<f16x4*>(o) = hi;
            *reinterpret_cast<f16x4*>(o + 32) = lo;
        } else {
            *reinterpret_cast<f32x4*>(of) = v;
        }
    }
    if constexpr (PLANES) pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax * A_SCALE < POPE_F16_OVERFLOW));
}

inline int grid_for(long long total) {
    const long long b = (total + 255) / 256;
    const long long cap = 64ll * pope_cu_count();
    return int(b < 1 ? 1 : (b < cap ? b : cap));
}

struct Plan {
    int n, H, W;
    int Hp[4], Wp[4];           // zero-bordered grids at 1/2, 1/4, 1/8 (index 1..3)
    size_t rows[4];
};
inline Plan make_plan(int n, int H, int W) {
    Plan p = {};
    p.n = n; p.H = H; p.W = W;
    for (int k = 1; k <= 3; ++k) {
        p.Hp[k] = (H >> k) + 2;
        p.Wp[k] = (W >> k) + 2;
        p.rows[k] = size_t(n) * p.Hp[k] * p.Wp[k];
    }
    return p;
}
inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

// workspace buffers, in floats-per-row (= bytes / 4 of a planes row or an fp32 row)
enum Buf { G0, P1A, P1B, P1C, F1, T1A, T1B, G2, D2, P2A, P2B, P2C, F2, T2A, T2B, X2O, G3, D3, P3A, P3B, P3C, NBUF };
struct BufSpec { int level, pitch; };
constexpr BufSpec kBufs[NBUF] = {
    {1, 64},  {1, 128}, {1, 128}, {1, 128}, {1, 224}, {1, 224}, {1, 224},
    {2, 1152}, {2, 128}, {2, 224}, {2, 224}, {2, 224}, {2, 256}, {2, 256}, {2, 256}, {2, 224},
    {3, 2016}, {3, 224}, {3, 256}, {3, 256}, {3, 256}};

}  // namespace

size_t pope_resnetfpn_workspace(int n, int H, int W) {
    const Plan p = make_plan(n, H, W);
    size_t total = 0;
    for (int b = 0; b < NBUF; ++b) total += align256(p.rows[kBufs[b].level] * kBufs[b].pitch * 4);
    return total;
}

int pope_launch_resnetfpn(const ResnetFpnParams& q, hipStream_t stream) {
    if (!q.img || !q.out_c || !q.out_f || !q.ws || q.n <= 0 || q.H < 16 || q.W < 16 || (q.H & 7) || (q.W & 7)) return POPE_ERR_ARG;
    const bool f32 = q.precision == POPE_PREC_F32_MFMA;
    if (!f32 && q.precision != POPE_PREC_F16X3) return POPE_ERR_ARG;
    for (int i = 0; i < 22; ++i)
        if (!(f32 ? static_cast<const void*>(q.wf[i]) : q.w[i])) return POPE_ERR_ARG;
    const Plan p = make_plan(q.n, q.H, q.W);
    for (int b = 0; b < NBUF; ++b)   // 32-bit buffer offsets everywhere: the caller splits larger batches
        if (p.rows[kBufs[b].level] * kBufs[b].pitch * 4ull >= (1ull << 32)) return POPE_ERR_ARG;
    if (q.ws_bytes < pope_resnetfpn_workspace(q.n, q.H, q.W)) return POPE_ERR_WORKSPACE;
    char* base = static_cast<char*>(q.ws);
    char* buf[NBUF];
    for (int b = 0; b < NBUF; ++b) {
        buf[b] = base;
        base += align256(p.rows[kBufs[b].level] * kBufs[b].pitch * 4);
    }
    // Borders and the channel padding (196 -> 224) must read as zeros.  Nothing is cleared wholesale: every planes
    // buffer is written in full by its producer — the GEMM epilogue zero-fills the padding columns, the gathers and
    // the FPN merge write zeros where they have no source — and has its border rows zeroed right after.

    int rc;
    // POPE_PREC_F32_MFMA (the range guard's re-run): the same buffers hold fp32 rows of the same pitch, every GEMM runs on
    // gemm_f32.hip (EPI_CONV, implicit 3x3 loader).  Its epilogue writes only the N real channels, so the channel padding
    // (196 -> 224) is cleared once up front; speed is not a goal of this path.
    if (f32 && hipMemsetAsync(q.ws, 0, pope_resnetfpn_workspace(q.n, q.H, q.W), stream) != hipSuccess) return POPE_ERR_LAUNCH;
    // generic GEMM over pixel rows: out[rows, N] = act(A[rows, K] . W^T + bias)
    auto gemm = [&](const void* a, int K, int wi, int N, int level, void* out_pl, float* out_f32, int out_pitch, float slope,
                    const void* res_pl, int res_pitch, bool conv, int a_pitch, const float* up_src = nullptr, int up_lds = 0) -> int {
        GemmParams g = {};
        const int Wp = p.Wp[level];
        const size_t shift = conv ? size_t(Wp) + 1 : 0;   // output (and shortcut) rows start at pixel Wp + 1
        if (f32) {
            g.A = static_cast<const float*>(a); g.W = q.wf[wi]; g.bias = q.b[wi];
            g.lda = a_pitch; g.ldw = K; g.ldc = out_pitch;
            g.M = int(p.rows[level] - (conv ? 2 * size_t(Wp) + 2 : 0)); g.N = N; g.K = K;
            g.epilogue = EPI_CONV;
            g.act_slope = slope;
            g.C = (out_pl ? static_cast<float*>(out_pl) : out_f32) + shift * out_pitch;
            if (res_pl) { g.res = static_cast<const float*>(res_pl) + shift * res_pitch; g.ldres = res_pitch; }
            if (conv) g.conv_wp = Wp;
            return pope_launch_gemm_nt_f32(g, stream);
        }
        g.a_pl = a; g.w_pl = q.w[wi]; g.bias = q.b[wi];
        g.lda = a_pitch; g.ldw = K; g.ldc = out_pitch;
        g.M = int(p.rows[level] - (conv ? 2 * size_t(Wp) + 2 : 0)); g.N = N; g.K = K;
        g.epilogue = EPI_CONV;
        g.act_slope = slope;
        if (out_pl) g.c_pl = static_cast<char*>(out_pl) + shift * out_pitch * 4;
        else g.C = out_f32 + shift * out_pitch;
        if (res_pl) { g.res_pl = static_cast<const char*>(res_pl) + shift * res_pitch * 4; g.ldres_pl = res_pitch; }
        if (conv) { g.conv_cch = a_pitch / 32; g.conv_wp = Wp; }
        if (up_src) {
            g.up_src = up_src; g.up_lds = up_lds; g.up_hp = p.Hp[level]; g.up_wp = Wp; g.up_n = q.n;
            const int H = g.up_hp - 2, W = g.up_wp - 2, Hs = H / 2, Ws = W / 2;
            g.up_sh = H > 1 ? float(Hs - 1) / float(H - 1) : 0.f;   // torch's area_pixel_compute_scale for align_corners
            g.up_sw = W > 1 ? float(Ws - 1) / float(W - 1) : 0.f;
            g.up_m_hw = unsigned((1ull << 32) / (unsigned long long)(g.up_hp) / (unsigned long long)(g.up_wp));
            g.up_m_w = unsigned((1ull << 32) / (unsigned long long)(g.up_wp));
        }
        g.range_flag = q.range_flag; g.range_bit = POPE_RANGE_INPUT;
        g.nbatch = 1;
        return pope_launch_planes16(g, stream);
    };
    auto zero_border = [&](void* b, int level, int pitch) -> int {
        const long long total = (long long)q.n * (2 * p.Wp[level] + 2 * (p.Hp[level] - 2)) * (pitch / 4);
        hipLaunchKernelGGL(zero_border_kernel, dim3(grid_for(total)), dim3(256), 0, stream, static_cast<u32x4*>(b), q.n,
                           p.Hp[level], p.Wp[level], pitch / 4);
        return pope_check_launch();
    };
    auto conv3 = [&](Buf in, int wi, int N, int level, Buf out, float slope, int res /* Buf or -1 */) -> int {
        const int ip = kBufs[in].pitch, op = kBufs[out].pitch;
        int r = gemm(buf[in], 9 * ip, wi, N, level, buf[out], nullptr, op, slope, res >= 0 ? buf[res] : nullptr,
                     res >= 0 ? kBufs[res].pitch : 0, true, ip);
        return r ? r : zero_border(buf[out], level, op);
    };
    auto gather = [&](Buf in, int level_in, Buf out, int taps) -> int {
        const int cch = kBufs[in].pitch / 32;
        const long long total = (long long)p.rows[level_in + 1] * taps * cch * 8;
        hipLaunchKernelGGL(gather_s2_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const u32x4*>(buf[in]),
                           reinterpret_cast<u32x4*>(buf[out]), q.n, p.Hp[level_in], p.Wp[level_in], cch, taps);
        return pope_check_launch();
    };
    // BasicBlock (resnet_fpn.py:15-40) with stride 1: x -> t -> y; returns with the block output in `y`
    auto block_s1 = [&](Buf x, Buf t, Buf y, int w0, int N, int level) -> int {
        int r = conv3(x, w0, N, level, t, 0.f, -1);
        return r ? r : conv3(t, w0 + 1, N, level, y, 0.f, x);
    };
    // BasicBlock with stride 2: x (level L) -> gathered taps -> t; shortcut 1x1 stride 2 -> s; y = relu(conv2(t) + s)
    // stride-2 convolution of x (level_in) -> out (level_in + 1) straight from x's planes (gemm_plain.hip, CONV = 2) where the call
    // fills a round of the CUs; returns 1 when the shape is not served (the caller gathers the taps instead: the same bits)
    auto conv_s2_implicit = [&](Buf x, int level_in, int taps, int wi, int N, Buf out, float slope) -> int {
        if (f32) return 1;
        const int lo = level_in + 1;
        GemmParams g = {};
        g.a_pl = buf[x]; g.w_pl = q.w[wi]; g.bias = q.b[wi];
        g.lda = kBufs[x].pitch; g.ldw = taps * kBufs[x].pitch; g.ldc = kBufs[out].pitch;
        g.M = int(p.rows[lo]); g.N = N; g.K = taps * kBufs[x].pitch;
        g.epilogue = EPI_CONV; g.act_slope = slope; g.c_pl = buf[out];
        g.conv_cch = kBufs[x].pitch / 32; g.conv_wp = p.Wp[level_in];
        g.conv_s2_taps = taps; g.conv_s2_hpi = p.Hp[level_in]; g.conv_s2_hpo = p.Hp[lo]; g.conv_s2_wpo = p.Wp[lo];
        g.conv_s2_in_rows = int(p.rows[level_in]);
        g.range_flag = q.range_flag; g.range_bit = POPE_RANGE_INPUT; g.nbatch = 1;
        if (!pope_wide_conv_s2_supported(g)) return 1;
        const int r = pope_launch_wide_conv_s2(g, stream);
        return r ? r : 0;
    };
    auto block_s2 = [&](Buf x, int level_in, Buf g9, Buf g1, Buf t, Buf s, Buf y, int w0, int N) -> int {
        const int lo = level_in + 1;
        int r = conv_s2_implicit(x, level_in, 9, w0, N, t, 0.f);
        if (r == 1) {
            r = gather(x, level_in, g9, 9);
            if (!r) r = gemm(buf[g9], kBufs[g9].pitch, w0, N, lo, buf[t], nullptr, kBufs[t].pitch, 0.f, nullptr, 0, false, kBufs[g9].pitch);
        }
        if (!r) r = zero_border(buf[t], lo, kBufs[t].pitch);
        if (r) return r;
        r = conv_s2_implicit(x, level_in, 1, w0 + 2, N, s, 1.f);
        if (r == 1) {
            r = gather(x, level_in, g1, 1);
            if (!r) r = gemm(buf[g1], kBufs[g1].pitch, w0 + 2, N, lo, buf[s], nullptr, kBufs[s].pitch, 1.f, nullptr, 0, false, kBufs[g1].pitch);
        }
        return r ? r : conv3(t, w0 + 1, N, lo, y, 0.f, s);
    };

    // stem (resnet_fpn.py:60-62,101)
    {
        const long long total = (long long)p.rows[1] * 8;
        if (f32) hipLaunchKernelGGL(stem_gather_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, q.img,
                                    reinterpret_cast<_Float16*>(buf[G0]), q.n, q.H, q.W, q.range_flag);
        else hipLaunchKernelGGL(stem_gather_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, q.img,
                                reinterpret_cast<_Float16*>(buf[G0]), q.n, q.H, q.W, q.range_flag);
        if ((rc = pope_check_launch())) return rc;
    }
    if ((rc = gemm(buf[G0], 64, 0, 128, 1, buf[P1A], nullptr, 128, 0.f, nullptr, 0, false, 64))) return rc;
    if ((rc = zero_border(buf[P1A], 1, 128))) return rc;
    // layer1 (1/2), layer2 (1/4), layer3 (1/8)
    if ((rc = block_s1(P1A, P1B, P1C, 1, 128, 1))) return rc;
    if ((rc = block_s1(P1C, P1A, P1B, 3, 128, 1))) return rc;             // x1 = P1B
    if ((rc = block_s2(P1B, 1, G2, D2, P2A, P2C, P2B, 5, 196))) return rc;
    if ((rc = block_s1(P2B, P2A, P2C, 8, 196, 2))) return rc;             // x2 = P2C
    if ((rc = block_s2(P2C, 2, G3, D3, P3A, P3C, P3B, 10, 256))) return rc;
    if ((rc = block_s1(P3B, P3A, P3C, 13, 256, 3))) return rc;            // x3 = P3C
    // FPN (resnet_fpn.py:107-117)
    if ((rc = gemm(buf[P3C], 256, 15, 256, 3, nullptr, q.out_c, 256, 1.f, nullptr, 0, false, 256))) return rc;             // x3_out
    // f16x3: the lateral 1 x 1 convolution's epilogue adds the bilinear x2 sample of the coarser map and writes the merged planes
    // (round 4: the fp32 lateral map and the upsample_add pass are gone — 1.9 GB of the call's fabric traffic at 48 images); the
    // fp32 mode keeps the two-step form
    if (!f32) {
        if ((rc = gemm(buf[P2C], 224, 16, 256, 2, buf[T2A], nullptr, 256, 1.f, nullptr, 0, false, 224, q.out_c, 256))) return rc;
        if ((rc = zero_border(buf[T2A], 2, 256))) return rc;
    } else {
    if ((rc = gemm(buf[P2C], 224, 16, 256, 2, nullptr, reinterpret_cast<float*>(buf[F2]), 256, 1.f, nullptr, 0, false, 224))) return rc;
    {
        const long long total = (long long)q.n * (p.Hp[2] - 2) * (p.Wp[2] - 2) * (256 / 4);
        if (f32) hipLaunchKernelGGL(upsample_add_planes_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const float*>(buf[F2]),
                                    256, q.out_c, 256, reinterpret_cast<_Float16*>(buf[T2A]), 256, 256, q.n, p.Hp[2], p.Wp[2], q.range_flag);
        else hipLaunchKernelGGL(upsample_add_planes_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const float*>(buf[F2]),
                                256, q.out_c, 256, reinterpret_cast<_Float16*>(buf[T2A]), 256, 256, q.n, p.Hp[2], p.Wp[2], q.range_flag);
        if ((rc = pope_check_launch())) return rc;
        if ((rc = zero_border(buf[T2A], 2, 256))) return rc;
    }
    }
    if ((rc = conv3(T2A, 17, 256, 2, T2B, 0.01f, -1))) return rc;
    if ((rc = gemm(buf[T2B], 9 * 256, 18, 196, 2, nullptr, reinterpret_cast<float*>(buf[X2O]), 224, 1.f, nullptr, 0, true, 256))) return rc;
    if (!f32) {
        if ((rc = gemm(buf[P1B], 128, 19, 196, 1, buf[T1A], nullptr, 224, 1.f, nullptr, 0, false, 128, reinterpret_cast<const float*>(buf[X2O]), 224))) return rc;
        if ((rc = zero_border(buf[T1A], 1, 224))) return rc;
    } else {
    if ((rc = gemm(buf[P1B], 128, 19, 196, 1, nullptr, reinterpret_cast<float*>(buf[F1]), 224, 1.f, nullptr, 0, false, 128))) return rc;
    {
        const long long total = (long long)q.n * (p.Hp[1] - 2) * (p.Wp[1] - 2) * (224 / 4);
        if (f32) hipLaunchKernelGGL(upsample_add_planes_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const float*>(buf[F1]),
                                    224, reinterpret_cast<const float*>(buf[X2O]), 224, reinterpret_cast<_Float16*>(buf[T1A]), 224, 196, q.n,
                                    p.Hp[1], p.Wp[1], q.range_flag);
        else hipLaunchKernelGGL(upsample_add_planes_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const float*>(buf[F1]),
                                224, reinterpret_cast<const float*>(buf[X2O]), 224, reinterpret_cast<_Float16*>(buf[T1A]), 224, 196, q.n,
                                p.Hp[1], p.Wp[1], q.range_flag);
        if ((rc = pope_check_launch())) return rc;
        if ((rc = zero_border(buf[T1A], 1, 224))) return rc;
    }
    }
    if ((rc = conv3(T1A, 20, 196, 1, T1B, 0.01f, -1))) return rc;
    return gemm(buf[T1B], 9 * 224, 21, 128, 1, nullptr, q.out_f, 128, 1.f, nullptr, 0, true, 224);                         // x1_out
}
