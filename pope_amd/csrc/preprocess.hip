// Batched image preprocessing on the GPU: what the reference does per proposal on the host with
// PIL + torchvision (segment_anything/segment_anything/dinov2_utils.py:55-78: ToPILImage -> Resize((256,256)) ->
// CenterCrop((196,196)) | Resize((224,224)) -> ToTensor -> Normalize) for P uint8 HWC crops at once, bit-identical to
// that host path: the resize is Pillow's 8-bit bilinear resample — two separable passes (horizontal, then vertical),
// per-output-pixel windows with 22-bit fixed-point weights, int32 accumulation from 2^21, each pass rounded to uint8 —
// with the window / weight tables built on the host in double precision exactly as Pillow builds them
// (pope_amd/preprocess.py:resize_tables); ToTensor and Normalize are one fp32 division and one fp32 subtract + division.
// Only the centre-crop window is computed.  Channel order is taken as given (the drivers pass cv2 BGR frames as RGB).
// HBM-bound streaming kernels: a thread per output pixel, the three channels of a pixel together, coalesced along x.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int PIL_BITS = 22;   // Pillow: PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PIL_BITS;
    return static_cast<unsigned char>(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass over the input rows the vertical pass will need, for the cropped output columns only:
// tmp[p][r][x][c], r = input row - row0, x = output column - left
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ img, int P, int Hin, int Win,
                                                        const int* __restrict__ hstart, const int* __restrict__ hcount,
                                                        const int* __restrict__ hk, int kh, int row0, int nrows, int left, int cw,
                                                        unsigned char* __restrict__ tmp) {
    const size_t total = size_t(P) * nrows * cw;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += size_t(gridDim.x) * 256) {
        const int x = int(i % cw);
        const size_t pr = i / cw;
        const int r = int(pr % nrows), p = int(pr / nrows);
        const int ox = left + x, s = hstart[ox], n = hcount[ox];
        const unsigned char* src = img + ((size_t(p) * Hin + row0 + r) * Win + s) * 3;
        const int* k = hk + size_t(ox) * kh;
        int a0 = 1 << (PIL_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
            a0 += int(src[3 * t]) * w;
            a1 += int(src[3 * t + 1]) * w;
            a2 += int(src[3 * t + 2]) * w;
        }
        unsigned char* o = tmp + i * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

// vertical pass + centre crop + ToTensor (/255) + Normalize ((v - mean) / std), NCHW fp32
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const unsigned char* __restrict__ tmp, int P, int row0, int nrows,
                                                             const int* __restrict__ vstart, const int* __restrict__ vcount,
                                                             const int* __restrict__ vk, int kv, int top, int ch, int cw,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             float* __restrict__ out) {
    const size_t total = size_t(P) * ch * cw;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += size_t(gridDim.x) * 256) {
        const int x = int(i % cw);
        const size_t py = i / cw;
        const int y = int(py % ch), p = int(py / ch);
        const int oy = top + y, s = vstart[oy], n = vcount[oy];
        const unsigned char* src = tmp + ((size_t(p) * nrows + (s - row0)) * cw + x) * 3;
        const int* k = vk + size_t(oy) * kv;
        int a0 = 1 << (PIL_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
            const unsigned char* q = src + size_t(t) * cw * 3;
            a0 += int(q[0]) * w;
            a1 += int(q[1]) * w;
            a2 += int(q[2]) * w;
        }
        const size_t plane = size_t(ch) * cw, o = size_t(p) * 3 * plane + size_t(y) * cw + x;
        out[o] = (float(clip8(a0)) / 255.0f - m0) / s0;
        out[o + plane] = (float(clip8(a1)) / 255.0f - m1) / s1;
        out[o + 2 * plane] = (float(clip8(a2)) / 255.0f - m2) / s2;
    }
}

// cv2.cvtColor(BGR2GRAY) for 8-bit images (OpenCV's fixed-point form: B 1868, G 9617, R 4899, 14 fractional bits,
// round to nearest) followed by `/ 255.` (eval_linemod_json.py:103-111): [P,H,W,3] uint8 BGR -> [P,1,H,W] fp32
__global__ __launch_bounds__(256) void gray_kernel(const unsigned char* __restrict__ bgr, size_t npix, float* __restrict__ out) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < npix; i += size_t(gridDim.x) * 256) {
        const unsigned char* q = bgr + i * 3;
        const int g = (int(q[0]) * 1868 + int(q[1]) * 9617 + int(q[2]) * 4899 + (1 << 13)) >> 14;
        out[i] = float(g) / 255.0f;
    }
}

// ToTensor + Normalize of a crop window, no resize (torchvision: x / 255, then (x - mean) / std, fp32): the dense
// pair path feeds 476 x 630 centre crops of 640 x 480 frames.  [P, Hin, Win, 3] uint8 -> [P, 3, ch, cw] fp32; a thread
// converts four pixels of a row (12 input bytes, three 16-byte stores when cw % 4 == 0)
__global__ __launch_bounds__(256) void crop_norm_kernel(const unsigned char* __restrict__ img, int P, int Hin, int Win, int top,
                                                        int left, int ch, int cw, float m0, float m1, float m2, float s0, float s1,
                                                        float s2, float* __restrict__ out) {
    const int qw = (cw + 3) / 4;
    const size_t total = size_t(P) * ch * qw, plane = size_t(ch) * cw;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += size_t(gridDim.x) * 256) {
        const int xq = int(i % qw);
        const size_t py = i / qw;
        const int y = int(py % ch), p = int(py / ch);
        const unsigned char* q = img + ((size_t(p) * Hin + top + y) * Win + left + 4 * xq) * 3;
        float* o = out + size_t(p) * 3 * plane + size_t(y) * cw + 4 * xq;
        const int nx = cw - 4 * xq < 4 ? cw - 4 * xq : 4;
        float r[3][4];
        for (int e = 0; e < nx; ++e) {
            r[0][e] = (float(q[3 * e]) / 255.0f - m0) / s0;
            r[1][e] = (float(q[3 * e + 1]) / 255.0f - m1) / s1;
            r[2][e] = (float(q[3 * e + 2]) / 255.0f - m2) / s2;
        }
        for (int c = 0; c < 3; ++c) {
            if (nx == 4 && !(cw & 3)) *reinterpret_cast<f32x4*>(o + c * plane) = f32x4{r[c][0], r[c][1], r[c][2], r[c][3]};
            else for (int e = 0; e < nx; ++e) o[c * plane + e] = r[c][e];
        }
    }
}


// cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) of uint8 images for P destination images at once — the drivers' proposal
// crops (utils/data_utils.py:239-255 called twice, eval_linemod_json.py:83-90).  minv[p][6] = the INVERSE map (destination
// pixel -> source position, what cv::warpAffine derives from the forward matrix) in fp64; evaluated per destination pixel
// the way OpenCV's 8-bit path does: 10-bit fixed point with a 1/64 px rounding offset, truncated to 1/32 px, the four
// neighbours blended with integer weights (32 - fx)(32 - fy) ... fx fy and rounded to nearest.  win[p] = (x0, y0, w, h):
// the source of image p is the w x h window of `img` whose top-left corner is (x0, y0) — pixels outside the window or
// outside `img` read as 0, which is exactly the intermediate zero-padded crop of the reference's first (integer
// translation) step, so the two warps of a proposal are ONE gather from the frame.
__global__ __launch_bounds__(256) void crop_warp_kernel(const unsigned char* __restrict__ img, int H, int W, int C,
                                                        const double* __restrict__ minv, const int* __restrict__ win, int P, int oh,
                                                        int ow, unsigned char* __restrict__ out) {
    const size_t total = size_t(P) * oh * ow;
    for (size_t id = size_t(blockIdx.x) * 256 + threadIdx.x; id < total; id += size_t(gridDim.x) * 256) {
        const int x = int(id % ow), y = int((id / ow) % oh), p = int(id / (size_t(ow) * oh));
        const double* m = minv + 6 * p;
        const int x0 = win[4 * p], y0 = win[4 * p + 1], ww = win[4 * p + 2], wh = win[4 * p + 3];
        // saturate_cast<int>(double) rounds to nearest even (rint)
        const long long adelta = (long long)rint(m[0] * x * 1024.0), bdelta = (long long)rint(m[3] * x * 1024.0);
        const long long X0 = (long long)rint((m[1] * y + m[2]) * 1024.0) + 16, Y0 = (long long)rint((m[4] * y + m[5]) * 1024.0) + 16;
        const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        const long long sx = X >> 5, sy = Y >> 5;
        const int fx = int(X & 31), fy = int(Y & 31);
        const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
        const bool okx0 = sx >= 0 && sx < ww && sx + x0 >= 0 && sx + x0 < W, okx1 = sx + 1 >= 0 && sx + 1 < ww && sx + 1 + x0 >= 0 && sx + 1 + x0 < W;
        const bool oky0 = sy >= 0 && sy < wh && sy + y0 >= 0 && sy + y0 < H, oky1 = sy + 1 >= 0 && sy + 1 < wh && sy + 1 + y0 >= 0 && sy + 1 + y0 < H;
        const unsigned char* r0 = img + (size_t(oky0 ? sy + y0 : 0) * W) * C;
        const unsigned char* r1 = img + (size_t(oky1 ? sy + 1 + y0 : 0) * W) * C;
        const size_t c0 = size_t(okx0 ? sx + x0 : 0) * C, c1 = size_t(okx1 ? sx + 1 + x0 : 0) * C;
        unsigned char* o = out + id * C;
        for (int c = 0; c < C; ++c) {
            const int p00 = (oky0 && okx0) ? r0[c0 + c] : 0, p01 = (oky0 && okx1) ? r0[c1 + c] : 0;
            const int p10 = (oky1 && okx0) ? r1[c0 + c] : 0, p11 = (oky1 && okx1) ? r1[c1 + c] : 0;
            o[c] = (unsigned char)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 512) >> 10);
        }
    }
}

inline unsigned grid_for(size_t n) { return unsigned(n / 256 + 1 < 16384 ? n / 256 + 1 : 16384); }

}  // namespace

int pope_launch_preprocess(const PreprocParams& p, hipStream_t stream) {
    if (!p.img || !p.out || !p.tmp || !p.hstart || !p.hcount || !p.hk || !p.vstart || !p.vcount || !p.vk) return POPE_ERR_ARG;
    if (p.P <= 0 || p.Hin <= 0 || p.Win <= 0 || p.ch <= 0 || p.cw <= 0 || p.top < 0 || p.left < 0 || p.kh <= 0 || p.kv <= 0) return POPE_ERR_ARG;
    if (p.row0 < 0 || p.nrows <= 0 || p.row0 + p.nrows > p.Hin) return POPE_ERR_ARG;
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for(size_t(p.P) * p.nrows * p.cw)), dim3(256), 0, stream, p.img, p.P, p.Hin, p.Win,
                       p.hstart, p.hcount, p.hk, p.kh, p.row0, p.nrows, p.left, p.cw, p.tmp);
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3(grid_for(size_t(p.P) * p.ch * p.cw)), dim3(256), 0, stream, p.tmp, p.P, p.row0,
                       p.nrows, p.vstart, p.vcount, p.vk, p.kv, p.top, p.ch, p.cw, p.mean[0], p.mean[1], p.mean[2], p.std[0],
                       p.std[1], p.std[2], p.out);
    return pope_check_launch();
}

int pope_launch_crop_norm(const unsigned char* img, int P, int Hin, int Win, int top, int left, int ch, int cw, const float* mean,
                          const float* std, float* out, hipStream_t stream) {
    if (!img || !out || !mean || !std || P <= 0 || ch <= 0 || cw <= 0 || top < 0 || left < 0 || top + ch > Hin || left + cw > Win)
        return POPE_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(out) & 15) return POPE_ERR_ARG;
    hipLaunchKernelGGL(crop_norm_kernel, dim3(grid_for(size_t(P) * ch * ((cw + 3) / 4))), dim3(256), 0, stream, img, P, Hin, Win, top,
                       left, ch, cw, mean[0], mean[1], mean[2], std[0], std[1], std[2], out);
    return pope_check_launch();
}

int pope_launch_gray(const unsigned char* bgr, size_t npix, float* out, hipStream_t stream) {
    if (!bgr || !out || !npix) return POPE_ERR_ARG;
    hipLaunchKernelGGL(gray_kernel, dim3(grid_for(npix)), dim3(256), 0, stream, bgr, npix, out);
    return pope_check_launch();
}

int pope_launch_crop_warp(const unsigned char* img, int H, int W, int C, const double* minv, const int* win, int P, int oh, int ow,
                          unsigned char* out, hipStream_t stream) {
    if (!img || !minv || !win || !out || H <= 0 || W <= 0 || C <= 0 || C > 4 || P <= 0 || oh <= 0 || ow <= 0) return POPE_ERR_ARG;
    hipLaunchKernelGGL(crop_warp_kernel, dim3(grid_for(size_t(P) * oh * ow)), dim3(256), 0, stream, img, H, W, C, minv, win, P, oh, ow, out);
    return pope_check_launch();
}
