// Row LayerNorm, fp32 (nn.LayerNorm(dim, eps=1e-6): dinov2/dinov2/models/vision_transformer.py:90;
// used as norm1/norm2 in block.py:56,68 and as the final norm, vision_transformer.py:230).
// HBM-bound streaming op: one wave per row, the row lives in registers (float2 per lane per
// 128 columns), two-pass mean / centred variance, wave-shuffle reductions, no LDS.
#include "common.h"
#include "kernels.h"

namespace {

template <int NV>  // dim = NV * 128
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y,
                                                         int ldy, int rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + size_t(row) * ldx;
    f32x2 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const f32x2*>(xr + i * 128 + lane * 2);
        s += v[i][0] + v[i][1];
    }
    constexpr float inv_d = 1.0f / float(NV * 128);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d0 = v[i][0] - mean, d1 = v[i][1] - mean;
        q += d0 * d0 + d1 * d1;
    }
    const float var = wave_sum(q) * inv_d;
    const float rstd = 1.0f / sqrtf(var + eps);
    float* yr = y + size_t(row) * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 128 + lane * 2;
        const f32x2 ww = *reinterpret_cast<const f32x2*>(w + c);
        const f32x2 bb = *reinterpret_cast<const f32x2*>(b + c);
        f32x2 o;
        o[0] = (v[i][0] - mean) * rstd * ww[0] + bb[0];
        o[1] = (v[i][1] - mean) * rstd * ww[1] + bb[1];
        *reinterpret_cast<f32x2*>(yr + c) = o;
    }
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Same LayerNorm, output as f16 hi/lo planes for the f16x3 GEMMs (y*scale = hi + lo): the split is
// done once here, in an HBM-bound kernel with idle VALU, instead of per K-step in every GEMM tile.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_planes_kernel(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ b,
                                                                _Float16* __restrict__ ypl,
                                                                int rows, float eps, float scale, unsigned* flag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + size_t(row) * ldx;
    f32x2 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const f32x2*>(xr + i * 128 + lane * 2);
        s += v[i][0] + v[i][1];
    }
    constexpr float inv_d = 1.0f / float(NV * 128);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d0 = v[i][0] - mean, d1 = v[i][1] - mean;
        q += d0 * d0 + d1 * d1;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
    _Float16* yr = ypl + size_t(row) * (2 * NV * 128);  // planes layout: per 32-column chunk [32 hi | 32 lo]
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 128 + lane * 2;
        const f32x2 ww = *reinterpret_cast<const f32x2*>(w + c);
        const f32x2 bb = *reinterpret_cast<const f32x2*>(b + c);
        f32x2 y;
        y[0] = ((v[i][0] - mean) * rstd * ww[0] + bb[0]) * scale;
        y[1] = ((v[i][1] - mean) * rstd * ww[1] + bb[1]) * scale;
        amax = pope_amax2(amax, y);
        const f16x2 hi = __builtin_convertvector(y, f16x2);
        const f16x2 lo = __builtin_convertvector(y - __builtin_convertvector(hi, f32x2), f16x2);
        *reinterpret_cast<f16x2*>(yr + (c >> 5) * 64 + (c & 31)) = hi;
        *reinterpret_cast<f16x2*>(yr + (c >> 5) * 64 + 32 + (c & 31)) = lo;
    }
    // a non-finite row (inf / NaN from an earlier overflow) has a non-finite mean or rstd; fmax ignores NaN
    pope_range_flag(flag, POPE_RANGE_LAYERNORM, !(amax < POPE_F16_OVERFLOW) || !(__builtin_fabsf(mean) + rstd < INFINITY));
}

// Same, two rows per wave (one per 32-lane half), 16-byte loads: dim = NV4 * 128 <= 512.  The 8-byte loads of the
// wave-per-row version reach 5.0 TB/s; a float4 copy reaches 6.3 on this chip.
typedef _Float16 f16x4_ln __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
template <int NV4>
__global__ __launch_bounds__(256) void layernorm_planes_half_kernel(const float* __restrict__ x, int ldx,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ b,
                                                                     _Float16* __restrict__ ypl, int rows, float eps,
                                                                     float scale, unsigned* flag) {
    const int l = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = row < rows;
    const float* xr = x + size_t(live ? row : 0) * ldx;
    f32x4 v[NV4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + i * 128 + l * 4);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    constexpr float inv_d = 1.0f / float(NV4 * 128);
    const float mean = half_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q += d * d;
        }
    const float rstd = 1.0f / sqrtf(half_sum(q) * inv_d + eps);
    if (!live) return;
    _Float16* yr = ypl + size_t(row) * (2 * NV4 * 128);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = i * 128 + l * 4;
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = ((v[i][e] - mean) * rstd * ww[e] + bb[e]) * scale;
        amax = pope_amax4(amax, y);
        const f16x4_ln hi = __builtin_convertvector(y, f16x4_ln);
        const f16x4_ln lo = __builtin_convertvector(y - __builtin_convertvector(hi, f32x4), f16x4_ln);
        *reinterpret_cast<f16x4_ln*>(yr + (c >> 5) * 64 + (c & 31)) = hi;
        *reinterpret_cast<f16x4_ln*>(yr + (c >> 5) * 64 + 32 + (c & 31)) = lo;
    }
    pope_range_flag(flag, POPE_RANGE_LAYERNORM, !(amax < POPE_F16_OVERFLOW) || !(__builtin_fabsf(mean) + rstd < INFINITY));
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, _Float16* __restrict__ pl,
                                                            size_t n2, int ld, float scale, unsigned* flag) {
    float amax = 0.f;
    bool nonfinite = false;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n2; i += size_t(gridDim.x) * 256) {
        const size_t e = 2 * i, row = e / ld;
        const int c = int(e - row * ld);
        const f32x2 y = *reinterpret_cast<const f32x2*>(src + e) * scale;
        amax = pope_amax2(amax, y);
        nonfinite |= !(y[0] + y[1] == y[0] + y[1]);
        const f16x2 h = __builtin_convertvector(y, f16x2);
        _Float16* o = pl + row * 2 * ld + (c >> 5) * 64 + (c & 31);
        *reinterpret_cast<f16x2*>(o) = h;
        *reinterpret_cast<f16x2*>(o + 32) = __builtin_convertvector(y - __builtin_convertvector(h, f32x2), f16x2);
    }
    pope_range_flag(flag, POPE_RANGE_INPUT, nonfinite || !(amax < POPE_F16_OVERFLOW));
}

__global__ __launch_bounds__(256) void div_planes_kernel(const float* __restrict__ src, long long bs, _Float16* __restrict__ pl,
                                                          int n, int rows, int cols, float divisor, float scale, unsigned* flag) {
    const size_t per = size_t(rows) * cols / 2, n2 = per * n;
    float amax = 0.f;
    bool nonfinite = false;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n2; i += size_t(gridDim.x) * 256) {
        const size_t b = i / per, e = (i - b * per) * 2, row = e / cols;
        const int c = int(e - row * cols);
        f32x2 y = *reinterpret_cast<const f32x2*>(src + b * bs + e);
        y[0] = (y[0] / divisor) * scale;   // the division first, exactly as the reference rounds it; scale is a power of two
        y[1] = (y[1] / divisor) * scale;
        amax = pope_amax2(amax, y);
        nonfinite |= !(y[0] + y[1] == y[0] + y[1]);
        const f16x2 hi = __builtin_convertvector(y, f16x2);
        _Float16* o = pl + (b * rows + row) * 2 * cols + (c >> 5) * 64 + (c & 31);
        *reinterpret_cast<f16x2*>(o) = hi;
        *reinterpret_cast<f16x2*>(o + 32) = __builtin_convertvector(y - __builtin_convertvector(hi, f32x2), f16x2);
    }
    pope_range_flag(flag, POPE_RANGE_MATCH, nonfinite || !(amax < POPE_F16_OVERFLOW));
}

// Patch embed operand: one thread per (row, k pair).  Row b*ntok + 1 + (py*gw + px), k = c*p*p + dy*p + dx reads
// img[b, c, py*p + dy, px*p + dx] (patch_embed.py:69-82: Conv2d with kernel = stride = patch, flattened row-major).
__global__ __launch_bounds__(256) void im2col_planes_kernel(const float* __restrict__ img, _Float16* __restrict__ pl,
                                                             int B, int H, int W, int patch, int kp, int ntok, int gw, unsigned* flag) {
    const size_t n2 = size_t(B) * ntok * (kp / 2);
    const int k_real = 3 * patch * patch, pp = patch * patch;
    float amax = 0.f;
    bool nonfinite = false;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n2; i += size_t(gridDim.x) * 256) {
        const size_t row = i / (kp / 2);
        const int k = int(i - row * (kp / 2)) * 2;
        const int b = int(row / ntok), t = int(row - size_t(b) * ntok);
        f32x2 v = {0.f, 0.f};
        if (t > 0) {
            const int py = (t - 1) / gw, px = (t - 1) - py * gw;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kk = k + e;
                if (kk < k_real) {
                    const int c = kk / pp, rem = kk - c * pp, dy = rem / patch, dx = rem - dy * patch;
                    v[e] = img[((size_t(b) * 3 + c) * H + py * patch + dy) * W + px * patch + dx];
                }
            }
        }
        v = v * K_PLANES_ACT_SCALE;
        amax = pope_amax2(amax, v);
        nonfinite |= !(v[0] + v[1] == v[0] + v[1]);
        const f16x2 hi = __builtin_convertvector(v, f16x2);
        _Float16* o = pl + row * 2 * kp + (k >> 5) * 64 + (k & 31);
        *reinterpret_cast<f16x2*>(o) = hi;
        *reinterpret_cast<f16x2*>(o + 32) = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);
    }
    pope_range_flag(flag, POPE_RANGE_PATCH, nonfinite || !(amax < POPE_F16_OVERFLOW));
}

// The same operand by strips: one workgroup per (image, patch row, group of IM2_P patches).  The 3 * patch image-row
// segments of the group are read once, coalesced, and scattered into LDS in OUTPUT order (per patch: k = c p^2 + dy p + dx);
// every thread then turns 8 consecutive k of one patch (two 16-byte LDS reads) into a 16-byte hi piece and its lo twin.  The
// per-(row, k) kernel above pays two divisions by run-time values per element (VALU-bound) and leaves neighbouring patches
// — which share every 128-byte line of the image — to workgroups on different XCDs (PMC fetch 3.2x the image).
constexpr int IM2_P = 16;
__global__ __launch_bounds__(256) void im2col_planes_strip_kernel(const float* __restrict__ img, _Float16* __restrict__ pl,
                                                                   int B, int H, int W, int patch, int kp, int ntok, int gw,
                                                                   int groups, unsigned* flag) {
    extern __shared__ __attribute__((aligned(16))) float seg[];   // [IM2_P][kp + 4]: the +4 floats skew the patches' banks
    const int gh = H / patch, ks = kp + 4;
    const int grp = blockIdx.x % groups, py = (blockIdx.x / groups) % gh, b = blockIdx.x / (groups * gh);
    const int px0 = grp * IM2_P, np = min(IM2_P, gw - px0);
    const int half_len = (np * patch) >> 1, nseg = 3 * patch, pp = patch * patch, k_real = 3 * pp;
    for (int i = threadIdx.x; i < np * (kp - k_real); i += 256) {        // padding columns read as zeros
        const int p = i / (kp - k_real);
        seg[p * ks + k_real + (i - p * (kp - k_real))] = 0.f;
    }
    // a segment is at most IM2_P * patch / 2 float2's: `lanes` threads per segment, 256 / lanes segments per pass
    const int lanes = half_len <= 32 ? 32 : half_len <= 64 ? 64 : half_len <= 128 ? 128 : 256;
    const int sub = threadIdx.x / lanes, j0 = threadIdx.x - sub * lanes, per_pass = 256 / lanes;
    for (int j = j0; j < half_len; j += lanes) {                          // one trip when half_len <= lanes
        const int p = (2 * j) / patch, dx = 2 * j - p * patch;            // patch is even: the pair stays inside one patch
        // segment sgi = c * patch + dy is image row (c H + py patch + dy): the row index advances by per_pass, plus
        // H - patch whenever dy wraps into the next channel (no division in the loop; eight loads in flight)
        const float* src = img + ((size_t(b) * 3 * H + size_t(py) * patch) * W + px0 * patch + 2 * j);
        float* dst = seg + p * ks + dx;
        int dy = sub % patch, c = sub / patch;
#pragma unroll 8
        for (int sgi = sub; sgi < nseg; sgi += per_pass) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(src + (size_t(c) * H + dy) * W);
            *reinterpret_cast<f32x2*>(dst + sgi * patch) = v;                 // c pp + dy patch == sgi patch
            dy += per_pass;
            if (dy >= patch) { dy -= patch; ++c; }
        }
    }
    __syncthreads();
    float amax = 0.f;
    bool nonfinite = false;
    const int pieces = kp >> 3;                                            // 8 k values per piece
    const size_t row0 = size_t(b) * ntok + 1 + size_t(py) * gw + px0;
    for (int i = threadIdx.x; i < np * pieces; i += 256) {
        const int p = i / pieces, k0 = (i - p * pieces) << 3;
        const f32x4 a = *reinterpret_cast<const f32x4*>(seg + p * ks + k0), c4 = *reinterpret_cast<const f32x4*>(seg + p * ks + k0 + 4);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (e < 4 ? a[e] : c4[e - 4]) * K_PLANES_ACT_SCALE;
            amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
            nonfinite |= !(v == v);
            hi[e] = _Float16(v);
            lo[e] = _Float16(v - float(hi[e]));
        }
        _Float16* o = pl + (row0 + p) * 2 * kp + (k0 >> 5) * 64 + (k0 & 31);
        *reinterpret_cast<f16x8*>(o) = hi;
        *reinterpret_cast<f16x8*>(o + 32) = lo;
    }
    if (py == 0 && grp == 0) {                           // the CLS row of the image: zeros (its value arrives with the pos table)
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        _Float16* o = pl + size_t(b) * ntok * 2 * kp;
        for (int i = threadIdx.x; i < 2 * kp / 8; i += 256) *reinterpret_cast<f16x8*>(o + 8 * i) = z;
    }
    pope_range_flag(flag, POPE_RANGE_PATCH, nonfinite || !(amax < POPE_F16_OVERFLOW));
}

__global__ __launch_bounds__(256) void range_check_kernel(const float* __restrict__ x, size_t n, float scale, unsigned* flag,
                                                           unsigned bit) {
    float amax = 0.f;
    bool nonfinite = false;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) {
        const float v = x[i] * scale;
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
        nonfinite |= !(v == v);
    }
    pope_range_flag(flag, bit, nonfinite || !(amax < POPE_F16_OVERFLOW));
}


// LayerNorm over 384 columns in EXACTLY the arithmetic of gemm_rowln.hip's fused epilogue — same association of every sum:
// per 16-lane row group, lane (wn, q4) adds its six column quads ((a0 + a1) + (a2 + a3)) in ascending order, the four q4
// lanes of a column wave combine as (s0 + s1) + (s2 + s3), the four column waves the same way; mean first, then the centred
// second moment; y = ((x - mean) * rstd) * w + b — so that a small batch, which runs the 128 x 128 residual GEMM + this kernel
// (the full-row-tile kernel cannot fill the chip below ~170 row tiles and serialises 12 / 48 K-steps per tile), returns
// bit for bit what the same image returns inside a large batch on the fused kernel.  16 rows per 256-thread block.
template <bool PLANES>
__global__ __launch_bounds__(256) void layernorm_rowln_order_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ b, void* __restrict__ out, int rows,
                                                                    float eps, unsigned* flag) {
    constexpr int RN = 384;
    const int j = threadIdx.x & 15, wn = j >> 2, q4 = j & 3;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < rows;
    const int col0 = wn * 96 + 4 * q4;
    f32x4 v[6];
    float s = 0.f;
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
        v[ni] = ok ? *reinterpret_cast<const f32x4*>(x + size_t(row) * RN + col0 + 16 * ni) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[ni][0] + v[ni][1]) + (v[ni][2] + v[ni][3]);
    }
    auto tree = [](float t) {   // q4 pairs, q4 halves, wave pairs, wave halves: commutative adds, fixed association
        t = t + __shfl_xor(t, 1);
        t = t + __shfl_xor(t, 2);
        t = t + __shfl_xor(t, 4);
        t = t + __shfl_xor(t, 8);
        return t;
    };
    const float mean = tree(s) * (1.0f / float(RN));
    float qs = 0.f;
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
        const f32x4 d = v[ni] - mean;
        v[ni] = d;
        qs += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    const float var = tree(qs) * (1.0f / float(RN));
    const float rstd = 1.0f / sqrtf(var + eps);
    f32x2 amax = {0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
        const int c = col0 + 16 * ni;
        const f32x4 lw = *reinterpret_cast<const f32x4*>(w + c), lb = *reinterpret_cast<const f32x4*>(b + c);
        const f32x4 y = v[ni] * rstd * lw + lb;
        if (!ok) continue;
        if constexpr (PLANES) {
            const f32x4 ys = y * K_PLANES_ACT_SCALE;
            pope_amax4x2(amax, ys);
            pope_f16x4 hi, lo;
            pope_split4(ys, hi, lo);
            _Float16* hp = static_cast<_Float16*>(out) + size_t(row) * 2 * RN + (c >> 5) * 64 + (c & 31);
            *reinterpret_cast<pope_f16x4*>(hp) = hi;
            *reinterpret_cast<pope_f16x4*>(hp + 32) = lo;
        } else {
            *reinterpret_cast<f32x4*>(static_cast<float*>(out) + size_t(row) * RN + c) = y;
        }
    }
    if constexpr (PLANES)
        pope_range_flag(flag, POPE_RANGE_LAYERNORM,
                        ok && (!(__builtin_fmaxf(amax[0], amax[1]) < POPE_F16_OVERFLOW) || !(__builtin_fabsf(mean) + rstd < INFINITY)));
}
}  // namespace

int pope_launch_range_check(const float* x, size_t n, float scale, unsigned* flag, unsigned bit, hipStream_t stream) {
    if (!x || !flag || !n) return POPE_ERR_ARG;
    const unsigned blocks = unsigned(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
    hipLaunchKernelGGL(range_check_kernel, dim3(blocks), dim3(256), 0, stream, x, n, scale, flag, bit);
    return pope_check_launch();
}

int pope_launch_div_planes(const float* src, long long bs, void* planes, int n, int rows, int cols, float divisor, float scale,
                           unsigned* flag, hipStream_t stream) {
    if (!src || !planes || n <= 0 || rows <= 0 || cols <= 0 || (cols & 31) || (bs & 1) || bs < (long long)rows * cols) return POPE_ERR_ARG;
    const size_t n2 = size_t(n) * rows * cols / 2;
    const unsigned blocks = unsigned(n2 / 256 + 1 < 65536 ? n2 / 256 + 1 : 65536);
    hipLaunchKernelGGL(div_planes_kernel, dim3(blocks), dim3(256), 0, stream, src, bs, static_cast<_Float16*>(planes), n, rows,
                       cols, divisor, scale, flag);
    return pope_check_launch();
}

int pope_launch_im2col_planes(const float* img, void* a_planes, int B, int H, int W, int patch, int kp, unsigned* flag,
                              hipStream_t stream) {
    if (!img || !a_planes || B <= 0 || patch <= 0 || H % patch || W % patch || (kp & 31) || kp < 3 * patch * patch) return POPE_ERR_ARG;
    const int gw = W / patch, ntok = 1 + (H / patch) * gw;
    const size_t n2 = size_t(B) * ntok * (kp / 2);
    const size_t strip_lds = size_t(IM2_P) * (kp + 4) * sizeof(float);
    const int groups = (gw + IM2_P - 1) / IM2_P;
    const long long strips = (long long)B * (H / patch) * groups;
    // patch >= 8: the strip kernel advances (dy, channel) by up to 8 segments per pass with ONE wrap (DINOv2: 14, SAM: 16);
    // smaller patches take the element-wise kernel below
    if (!(patch & 1) && patch >= 8 && strip_lds <= 48 * 1024 && strips < (1ll << 31) && !(reinterpret_cast<uintptr_t>(img) & 7) &&
        !(reinterpret_cast<uintptr_t>(a_planes) & 15)) {
        hipLaunchKernelGGL(im2col_planes_strip_kernel, dim3(unsigned(strips)), dim3(256), strip_lds, stream, img,
                           static_cast<_Float16*>(a_planes), B, H, W, patch, kp, ntok, gw, groups, flag);
        return pope_check_launch();
    }
    const unsigned blocks = unsigned(n2 / 256 + 1 < 65536 ? n2 / 256 + 1 : 65536);
    hipLaunchKernelGGL(im2col_planes_kernel, dim3(blocks), dim3(256), 0, stream, img, static_cast<_Float16*>(a_planes), B, H, W,
                       patch, kp, ntok, gw, flag);
    return pope_check_launch();
}

int pope_launch_layernorm_planes(const float* x, int ldx, const float* w, const float* b, void* y_pl,
                                 int rows, int dim, float eps, unsigned* flag, hipStream_t stream) {
    if (rows <= 0 || dim <= 0 || (dim & 127) || dim > 2048 || (ldx & 1) || !y_pl) return POPE_ERR_ARG;
    _Float16* ypl = static_cast<_Float16*>(y_pl);
    if (dim <= 512 && !(ldx & 3) && !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(w) & 15) &&
        !(reinterpret_cast<uintptr_t>(b) & 15)) {  // two rows per wave, 16-byte accesses
        const dim3 g8((rows + 7) / 8), b256(256);
        switch (dim / 128) {
            case 1: hipLaunchKernelGGL(layernorm_planes_half_kernel<1>, g8, b256, 0, stream, x, ldx, w, b, ypl, rows, eps, K_PLANES_ACT_SCALE, flag); break;
            case 2: hipLaunchKernelGGL(layernorm_planes_half_kernel<2>, g8, b256, 0, stream, x, ldx, w, b, ypl, rows, eps, K_PLANES_ACT_SCALE, flag); break;
            case 3: hipLaunchKernelGGL(layernorm_planes_half_kernel<3>, g8, b256, 0, stream, x, ldx, w, b, ypl, rows, eps, K_PLANES_ACT_SCALE, flag); break;
            default: hipLaunchKernelGGL(layernorm_planes_half_kernel<4>, g8, b256, 0, stream, x, ldx, w, b, ypl, rows, eps, K_PLANES_ACT_SCALE, flag); break;
        }
        return pope_check_launch();
    }
    const dim3 grid((rows + 3) / 4), block(256);
#define POPE_LNP_CASE(NV)                                                                                    \
    case NV:                                                                                                 \
        hipLaunchKernelGGL(layernorm_planes_kernel<NV>, grid, block, 0, stream, x, ldx, w, b, ypl, rows, eps,    \
                           K_PLANES_ACT_SCALE, flag);                                                     \
        break;
    switch (dim / 128) {
        POPE_LNP_CASE(1) POPE_LNP_CASE(2) POPE_LNP_CASE(3) POPE_LNP_CASE(4) POPE_LNP_CASE(5) POPE_LNP_CASE(6) POPE_LNP_CASE(8)
        POPE_LNP_CASE(10) POPE_LNP_CASE(12) POPE_LNP_CASE(16)   // 5, 10: the SAM ViT-H width 1280 and its test twin 640
        default: return POPE_ERR_ARG;
    }
#undef POPE_LNP_CASE
    return pope_check_launch();
}

int pope_launch_split_planes(const float* src, void* pl, int rows, int ld, float scale, unsigned* flag, hipStream_t stream) {
    if (!src || !pl || rows <= 0 || ld <= 0 || (ld & 31)) return POPE_ERR_ARG;
    const size_t n2 = size_t(rows) * ld / 2;
    const unsigned blocks = unsigned(n2 / 256 + 1 < 4096 ? n2 / 256 + 1 : 4096);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, stream, src, static_cast<_Float16*>(pl), n2, ld, scale, flag);
    return pope_check_launch();
}

int pope_launch_layernorm_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                              int rows, int dim, float eps, hipStream_t stream) {
    if (rows <= 0 || dim <= 0 || (dim & 127) || dim > 2048 || (ldx & 1) || (ldy & 1)) return POPE_ERR_ARG;
    const dim3 grid((rows + 3) / 4), block(256);
#define POPE_LN_CASE(NV)                                                                              \
    case NV:                                                                                          \
        hipLaunchKernelGGL(layernorm_kernel<NV>, grid, block, 0, stream, x, ldx, w, b, y, ldy, rows, eps); \
        break;
    switch (dim / 128) {
        POPE_LN_CASE(1) POPE_LN_CASE(2) POPE_LN_CASE(3) POPE_LN_CASE(4) POPE_LN_CASE(5) POPE_LN_CASE(6) POPE_LN_CASE(8)
        POPE_LN_CASE(10) POPE_LN_CASE(12) POPE_LN_CASE(16)   // 5, 10: the SAM ViT-H width 1280 and its test twin 640
        default: return POPE_ERR_ARG;
    }
#undef POPE_LN_CASE
    return pope_check_launch();
}

int pope_launch_layernorm_rowln_order(const float* x, const float* w, const float* b, void* y_planes, float* y_f32, int rows, float eps,
                                      unsigned* flag, hipStream_t stream) {
    if (!x || !w || !b || (!y_planes) == (!y_f32) || rows <= 0) return POPE_ERR_ARG;
    const dim3 grid((rows + 15) / 16), block(256);
    if (y_planes) hipLaunchKernelGGL(layernorm_rowln_order_kernel<true>, grid, block, 0, stream, x, w, b, y_planes, rows, eps, flag);
    else hipLaunchKernelGGL(layernorm_rowln_order_kernel<false>, grid, block, 0, stream, x, w, b, static_cast<void*>(y_f32), rows, eps, flag);
    return pope_check_launch();
}
