// Internal launcher interface between the kernel translation units and the C ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/pope_hip.h"  // POPE_RANGE_* bits of the f16x3 range-guard word

enum GemmEpilogue {
    EPI_BIAS = 0,         // C = A.W^T + bias
    EPI_BIAS_GELU = 1,    // C = gelu_erf(A.W^T + bias)
    EPI_BIAS_LS_RES = 2,  // C = res + gamma * (A.W^T + bias)      (LayerScale + residual)
    EPI_POSB = 3,         // patch embed: C = im2col(img).W^T + posb[token]   (A = image)
    EPI_BIAS_RELU = 5,    // C = max(A.W^T + bias, 0)                (LoFTR encoder MLP, transformer.py:24-28; gemm_planes.hip only)
    EPI_SIM = 4,          // batched similarity (planes kernel): C[b] = (A[b].W[b]^T * alpha) / divisor, no bias
    EPI_SAM_QKV = 7,      // SAM block's QKV projection written straight into the attention operand planes (sam.hip; gemm_planes.hip only)
    EPI_QKV_F16 = 8,      // plain f16 only (POPE_PREC_F16 ViT blocks): C = (A.W^T + bias) * (col < sam_dim ? sam_qscale : 1) -> f16 row-major
                          // [M, N] with NO activation scale: the operand of attention_f16.hip (q carries head_dim^-0.5 * log2 e before its ONE rounding)
    EPI_CONV_UP = 9,      // kernel-side instantiation of EPI_CONV with GemmParams::up_src (callers pass EPI_CONV)
    EPI_CONV = 6,         // C = act(A.W^T + bias [+ res_pl]), act(v) = max(v, 0) + act_slope * min(v, 0): ReLU (0), LeakyReLU
                          // (0.01) or identity (1); the ResNet-FPN convolutions (conv.hip; gemm_planes.hip only)
};

struct GemmParams {
    const float* A;     // [M, lda] row-major activations (EPI_POSB: image [B,3,H,W])
    const float* W;     // [N, ldw] row-major weights (torch Linear layout, out x in)
    const float* bias;  // [N] or null
    float* C;           // [M, ldc]
    int lda, ldw, ldc;
    int M, N, K;
    int epilogue;
    const float* gamma;  // [N]      (EPI_BIAS_LS_RES)
    const float* res;    // [M,ldres](EPI_BIAS_LS_RES; may alias C)
    int ldres;
    // EPI_SIM: nbatch problems of [rows_a x rows_w x K]; A planes [nbatch*rows_a, lda], W planes [nbatch*rows_w, ldw],
    // C [nbatch, rows_a, rows_w] fp32 (ldc = rows_w); M = rows_a, N = rows_w
    int nbatch;
    float alpha, divisor;
    // EPI_SIM fused softmax statistics (dense matcher, match.hip): every wave writes, for its 64-column block, the
    // (max, sum exp(. - max)) of each of its rows, and for each of its two 32-row blocks the same per column:
    //   row_part  [nbatch][M][ncb]      float2, ncb = 2 * ceil(N / 128)
    //   col_pmax / col_psum [nbatch][nrb][ldp] floats, nrb = 4 * ceil(M / 128), ldp = N rounded up to 4
    // divisor_eff = divisor / alpha and rdiv = 1 / divisor_eff (host, correctly rounded): sim = acc / divisor_eff
    float* row_part;
    float* col_pmax;
    float* col_psum;
    int ncb, nrb, ldp;
    float divisor_eff, rdiv;
    int res_mod;         // > 0: the residual row is (row % res_mod) of a [res_mod, ldres] table (patch embed: pos + bias)
    // f16x3 "planes" operands (gemm_planes.hip): a tensor X[rows, ld] kept as two f16
    // planes with X = (hi + lo) / scale (power-of-two scale: activations 8, weights 256).  The planes
    // kernel needs a_pl and w_pl; when c_pl is set its epilogue writes planes (scale 8) instead of fp32 C.
    // Layout of a planes tensor [rows, ld]: row-major, 2*ld halves per row, per 32-column chunk the
    // 32 hi halves then the 32 lo halves (128 contiguous bytes = one cache line per row and K-step):
    //   offset(row, col, plane) = row*2*ld + (col>>5)*64 + plane*32 + (col&31).
    const void* a_pl;
    const void* w_pl;
    void* c_pl;
    // f16x3 range guard: a planes-writing epilogue ORs range_bit into *range_flag (may be null) when a value does
    // not fit f16 after scaling; the on-the-fly f16x3 kernel checks its operands before the split
    unsigned* range_flag;
    unsigned range_bit;
    // fused following LayerNorm (gemm_rowln.hip, N = 384): x = res + gamma * (A.W^T + bias) -> C, and
    // LayerNorm(x) * ln_w + ln_b -> ln_planes (activation planes) or ln_f32 (exactly one of the two)
    const float* ln_w; const float* ln_b;
    float ln_eps;
    void* ln_planes;
    float* ln_f32;
    int rl_prefetch;     // gemm_rowln.hip: touch the tile's residual lines during its K loop (set by the launcher)
    // EPI_CONV (gemm_planes.hip): optional residual as activation planes [M, ldres_pl] (BasicBlock shortcut), activation
    // slope, and — conv_cch > 0 — the implicit 3x3 stride-1 convolution over a zero-bordered NHWC planes tensor
    // [n, Hp, Wp = conv_wp, 32 * conv_cch channels]: output row R = pixel R + Wp + 1, and the A row of K-step
    // (tap (dy, dx), chunk c) is row R + dy * Wp + dx, chunk c — a row-uniform shift, i.e. a scalar offset per K-step
    // (K = 9 * 32 * conv_cch; W rows hold the taps in (dy, dx, channel) order); a_pl may be read up to 2 Wp + 2 rows
    // past M (the caller's buffer has them, or the range check returns zeros)
    const void* res_pl;
    int ldres_pl;
    float act_slope;
    int conv_cch, conv_wp;
    // gemm_plain.hip CONV = 2 (pope_launch_wide_conv_s2): stride-2 convolution read straight from the zero-bordered INPUT tensor
    // [n, conv_s2_hpi, conv_wp, 32 * conv_cch] (conv_s2_in_rows pixel rows) for the output grid [n, conv_s2_hpo, conv_s2_wpo];
    // conv_s2_taps = 9 (3 x 3, pad 1) or 1 (the 1 x 1 shortcut)
    int conv_s2_taps, conv_s2_hpi, conv_s2_hpo, conv_s2_wpo, conv_s2_in_rows;
    // EPI_CONV with up_src (gemm_planes.hip; round 4): the FPN merge of resnet_fpn.py:109-115 in the lateral 1 x 1 convolution's
    // epilogue — output row R is pixel R of this level's zero-bordered grid [up_n, up_hp, up_wp]; the bilinear x2
    // (align_corners) sample of the half-resolution fp32 map up_src [up_n, (up_hp - 2) / 2 + 2, (up_wp - 2) / 2 + 2, up_lds] at
    // that pixel is added before the activation.  Replaces conv.hip's upsample_add pass (same arithmetic, same bits) and the
    // fp32 round trip of the lateral map.  Not together with res_pl.
    const float* up_src;
    int up_lds, up_hp, up_wp, up_n;
    float up_sh, up_sw;          // bilinear scales (Hs - 1) / (H - 1), (Ws - 1) / (W - 1) (0 for a single output row / column)
    unsigned up_m_hw, up_m_w;    // floor(2^32 / (up_hp * up_wp)), floor(2^32 / up_wp)
    // plain != 0 (gemm_planes.hip only): single-product f16 arithmetic.  a_pl / w_pl (and c_pl) are f16 ROW-MAJOR tensors
    // (value * scale); K, lda, ldw and a planes output's ldc are given in 64-bit column PAIRS (= real columns / 2, so
    // that a row's pitch is still ld * 4 bytes); N, ldc of an fp32 output, bias, gamma and res stay in real columns.
    int plain;
    // EPI_SAM_QKV: C[t, which * dim + head * hd + c] (+ bias, q additionally * sam_qscale) goes to row
    // sam_rowmap[t] + head * sam_npad, column c of sam_q / sam_k / sam_v (f16 hi/lo rows [DQ | DQ], [DQ | hd], [DV | DV];
    // plain: the hi parts only) — the window partition of image_encoder.py:238-259 is that row map
    void *sam_q, *sam_k, *sam_v;
    unsigned sam_bytes[3];
    const int* sam_rowmap;
    int sam_hd, sam_dim, sam_npad, sam_dq, sam_dv;
    float sam_qscale;
    // patch-embed gather (EPI_POSB)
    const float* posb;   // [ntok, N]: row 0 = cls_token + pos[0]; row n = conv bias + pos[n]
    int ntok, img_h, img_w, patch, grid_w;
};

int pope_launch_gemm_nt_f32(const GemmParams& g, hipStream_t stream);

// Same contract on the f16 matrix cores with error-compensated operands (gemm_f16x3.hip).
bool pope_gemm_f16x3_supported(const GemmParams& g);
// Planes variant: W (and optionally A / C) as pre-split f16 planes, no splitting in the K loop.
int pope_launch_gemm_nt_f16x3_planes(const GemmParams& g, hipStream_t stream);
int pope_launch_sim_f16x3_planes(const GemmParams& g, hipStream_t stream);  // EPI_SIM, batched
// residual GEMM + the following LayerNorm in one kernel (gemm_rowln.hip; N = 384 only: `supported` says)
bool pope_gemm_rowln_supported(const GemmParams& g);
int pope_launch_gemm_rowln(const GemmParams& g, hipStream_t stream);
// the same 192 x 384 LDS-direct tile stream for planes -> planes Linears whose width is a multiple of 384 (gemm_rowln.hip)
bool pope_stream384_supported(const GemmParams& g);
int pope_launch_stream384(const GemmParams& g, hipStream_t stream);
// plain-f16 long-K mainloop (gemm_plain.hip: 256-row tiles, LDS-direct staging) for the GemmParams::plain shapes it serves;
// same results as pope_launch_planes16 on them
bool pope_wide_x3_supported(const GemmParams& g);   // the same mainloop on f16x3 planes -> planes (BIAS, BIAS_GELU) at large M
int pope_launch_wide_x3(const GemmParams& g, hipStream_t stream);
bool pope_wide_conv_s2_supported(const GemmParams& g);   // stride-2 convolutions without the gathered tap tensor (large M)
int pope_launch_wide_conv_s2(const GemmParams& g, hipStream_t stream);
bool pope_wide_conv_supported(const GemmParams& g);  // the implicit 3 x 3 convolutions (EPI_CONV, conv_cch > 0) at large M
int pope_launch_wide_conv(const GemmParams& g, hipStream_t stream);
bool pope_plain256_supported(const GemmParams& g);
int pope_launch_plain256(const GemmParams& g, hipStream_t stream);
// the v_mfma_f32_16x16x32_f16 mainloop (gemm_planes.hip) behind both of the above; arguments already validated
int pope_launch_planes16(const GemmParams& g, hipStream_t stream);
constexpr float K_PLANES_ACT_SCALE = 8.0f, K_PLANES_W_SCALE = 256.0f;  // == POPE_PLANES_*_SCALE of pope_hip.h

// y = LayerNorm(x) written as f16 planes (scale POPE_PLANES_ACT_SCALE), [rows, dim] halves each.
int pope_launch_layernorm_planes(const float* x, int ldx, const float* w, const float* b, void* y_pl,
                                 int rows, int dim, float eps, unsigned* range_flag, hipStream_t stream);
// LayerNorm(384) in the exact arithmetic of gemm_rowln.hip's fused epilogue (bit-identical results): the small-batch twin
int pope_launch_layernorm_rowln_order(const float* x, const float* w, const float* b, void* y_planes, float* y_f32, int rows, float eps,
                                      unsigned* flag, hipStream_t stream);
// Generic fp32 [rows, ld] -> planes converter (ld % 32 == 0).
// patch embed, f16x3: image [B,3,H,W] -> A planes [B*ntok, kp] (kp = 3*patch^2 rounded up to 32; row b*ntok is the
// all-zero CLS row, row b*ntok + 1 + n the flattened patch n; zero K padding)
int pope_launch_im2col_planes(const float* img, void* a_planes, int B, int H, int W, int patch, int kp, unsigned* range_flag,
                              hipStream_t stream);
// matcher operand: planes of feat / divisor (a true fp32 division, coarse_matching.py:109) for n blocks of `rows` rows
// with `bs` elements between blocks; output compact [n*rows, cols]
int pope_launch_div_planes(const float* src, long long bs, void* planes, int n, int rows, int cols, float divisor, float scale,
                           unsigned* range_flag, hipStream_t stream);
int pope_launch_split_planes(const float* src, void* pl, int rows, int ld, float scale, unsigned* range_flag, hipStream_t stream);
// HBM-bound scan: ORs `bit` into *flag when any |x[i]| * scale >= 65520 or x[i] is not finite (operand check of the
// op-level f16x3 entry points, which split fp32 operands inside their K loops)
int pope_launch_range_check(const float* x, size_t n, float scale, unsigned* flag, unsigned bit, hipStream_t stream);
int pope_launch_gemm_nt_f16x3(const GemmParams& g, hipStream_t stream);

// y[r,:] = LayerNorm(x[r,:]) * w + b over `dim` (multiple of 128, <= 2048), eps inside the sqrt.
int pope_launch_layernorm_f32(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                              int rows, int dim, float eps, hipStream_t stream);

// POPE_PREC_F16 pieces of the DINOv2 path (sam.hip, attention_f16.hip): LayerNorm -> f16 row-major (value * 8); attention on the f16
// operands of the QKV epilogue, f16 row-major output (value * 8)
int pope_launch_layernorm_f16(const float* x, const float* w, const float* b, void* y_f16, int rows, int dim, float eps, unsigned* flag,
                              hipStream_t stream);
// round 4: f16 qkv in (EPI_QKV_F16), LDS-direct K / V staging, four stages, software-pipelined (attention_f16.hip)
int pope_launch_attention_f16_dma(const void* qkv_f16, void* out_f16, int B, int N, int heads, hipStream_t stream);
// Multi-head softmax attention over qkv[B, N, 3, heads, 64] -> out[B, N, heads*64].
int pope_launch_attention_f32(const float* qkv, float* out, int B, int N, int heads, hipStream_t stream);
int pope_launch_attention_f16x3(const float* qkv, float* out, int B, int N, int heads, hipStream_t stream);
// same, output as activation planes [B*N, heads*64] for the f16x3 proj GEMM
int pope_launch_attention_f16x3_planes(const float* qkv, void* out_planes, int B, int N, int heads, hipStream_t stream);
// qkv given as planes (the QKV GEMM epilogue's output), output planes: the whole-model f16x3 dataflow
int pope_launch_attention_f16x3_planes_io_diag(const void* qkv_planes, void* out_planes, int B, int N, int heads, long long* exact_passes_host,
                                               hipStream_t stream);   // diagnostic twin: counts exact passes, synchronises
int pope_launch_attention_f16x3_planes_io(const void* qkv_planes, void* out_planes, int B, int N, int heads, hipStream_t stream);

struct MatchParams {
    const float* feat0;  // [n, L, C]
    const float* feat1;  // [n, S, C]
    int n, L, S, C;
    long long bs0, bs1;  // elements between consecutive pairs in feat0 / feat1 (>= L*C, S*C)
    int h0, w0, h1, w1;  // coarse grids (L = h0*w0, S = h1*w1)
    float thr, temperature;
    int border;
    float scale;         // hw0_i[0] / hw0_c[0]
    float* sim;          // [n, L, S]: sim; conf in place when publish_conf (then it is the caller's conf_matrix)
    int publish_conf;
    void* planes0; void* planes1;  // optional scratch [n*L*C] / [n*S*C] floats-worth: similarity on the f16 matrix cores
    // pieces of the softmax statistics written by the f16x3 contraction's epilogue (GemmParams: row_part, col_pmax, col_psum)
    float* row_part; float* col_pmax; float* col_psum;
    int ncb, nrb, ldp;               // 2 * ceil(S/128), 4 * ceil(L/128), S rounded up to 4
    float* row_max; float* row_sum;  // [n, L]   softmax(sim, dim=2) statistics
    float* col_max; float* col_sum;  // [n, S]   softmax(sim, dim=1) statistics
    float* colmax_part;              // [n, nrb2, ldp] column maxima of conf per 32-row block
    int nrb2;                        // pope_match_nrb2(L)
    float* conf_colmax;              // [n, S]
    float* conf_rowmax;              // [n, L]
    int* row_arg; int* row_cnt;      // [n, L]  first argmax column of conf, number of columns attaining the maximum
    int* row_j;                      // [n, L]  matched column or -1
    float* row_conf;                 // [n, L]
    unsigned* range_flag;            // optional f16x3 range-guard word (POPE_RANGE_MATCH)
    int* counts;                     // [n + 1] per-pair match counts, then total
    // compacted outputs (capacity n * L)
    long long* b_ids; long long* i_ids; long long* j_ids;
    float* mconf; float* mkpts0; float* mkpts1;
};
int pope_match_nrb2(int L);
int pope_launch_dense_match_f32(const MatchParams& p, hipStream_t stream);

// Batched PIL-exact preprocessing (preprocess.hip): resize (window / fixed-point weight tables from the host) ->
// centre crop -> /255 -> normalise, uint8 HWC [P, Hin, Win, 3] -> fp32 NCHW [P, 3, ch, cw]
struct PreprocParams {
    const unsigned char* img;
    int P, Hin, Win;
    const int *hstart, *hcount, *hk; int kh;   // horizontal tables, indexed by OUTPUT column of the full resized image
    const int *vstart, *vcount, *vk; int kv;   // vertical tables, indexed by output row
    int top, left, ch, cw;                     // crop window in the resized image
    int row0, nrows;                           // input rows the cropped output rows read: [row0, row0 + nrows)
    float mean[3], std[3];
    unsigned char* tmp;                        // [P, nrows, cw, 3]
    float* out;
};
int pope_launch_preprocess(const PreprocParams& p, hipStream_t stream);
int pope_launch_gray(const unsigned char* bgr, size_t npix, float* out, hipStream_t stream);
int pope_launch_crop_warp(const unsigned char* img, int H, int W, int C, const double* minv, const int* win, int P, int oh, int ow,
                          unsigned char* out, hipStream_t stream);
int pope_launch_crop_norm(const unsigned char* img, int P, int Hin, int Win, int top, int left, int ch, int cw, const float* mean,
                          const float* std, float* out, hipStream_t stream);

// One LoFTR encoder layer update (loftr.hip): x <- layer(x, source); weights as f16x3 planes (bias-free Linears)
struct LoftrLayerParams {
    float* x;              // [n, L, C] in / out
    const float* source;   // [n, S, C] (may be x itself: 'self' layers)
    int n, L, S, C, H;
    const void *q_wp, *kv_wp, *merge_wp, *mlp0_wp, *mlp1_wp;   // [C,C], [2C,C] (k rows then v rows), [C,C], [2C,2C], [C,2C]
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
    const float *q_w, *kv_w, *merge_w, *mlp0_w, *mlp1_w;       // POPE_PREC_F32_MFMA: the same matrices as fp32
    int precision;                                             // POPE_PREC_F16X3 | POPE_PREC_F32_MFMA
    float ln_eps;
    void* ws; size_t ws_bytes;
    unsigned* range_flag;
};
// ResNet-FPN local-feature CNN of the LoFTR matcher (conv.hip; resnet_fpn.py:43-118)
struct ResnetFpnParams {
    const float* img;          // [n, 1, H, W] gray in [0, 1]
    int n, H, W;               // H, W multiples of 8
    const void* w[22];         // weight planes (scale 256), BatchNorm folded: order in pope_hip.h
    const float* b[22];        // folded biases (null where the reference has neither bias nor BatchNorm)
    const float* wf[22];       // POPE_PREC_F32_MFMA: the same [Cout, K] matrices as fp32 (w may then be null)
    int precision;             // POPE_PREC_F16X3 | POPE_PREC_F32_MFMA
    float* out_c;              // [n, H/8 + 2, W/8 + 2, 256] fp32, zero border
    float* out_f;              // [n, H/2 + 2, W/2 + 2, 128] fp32, border undefined
    void* ws; size_t ws_bytes;
    unsigned* range_flag;
};
size_t pope_resnetfpn_workspace(int n, int H, int W);
int pope_launch_resnetfpn(const ResnetFpnParams& q, hipStream_t stream);
// LoFTR fine stage (fine.hip; fine_preprocess.py:29-59, fine_matching.py:15-74)
struct FinePreParams {
    const float *f0, *f1;          // 1/2-resolution maps, addressed through element strides (n, c, h, w)
    long long s0[4], s1[4];
    int H0, W0, H1, W1;            // map sizes
    int wc0, wc1;                  // coarse grid widths (cell id = cy * wc + cx)
    const float *fc0, *fc1;        // [n, L, Cc], [n, S, Cc] coarse features after the coarse transformer
    int L, S, Cc, Cf;
    const long long *b_ids, *i_ids, *j_ids;   // [M] matches (device)
    int M, Wn, stride;             // window size (5), fine pixels per coarse cell
    const void *down_wp, *merge_wp;   // [Cf, Cc], [Cf, 2 Cf] weight planes
    const float *down_b, *merge_b;
    const float *down_w, *merge_w;    // POPE_PREC_F32_MFMA: the same matrices as fp32
    int precision;                    // POPE_PREC_F16X3 | POPE_PREC_F32_MFMA
    float* out;                    // [2 M, Wn * Wn, Cf]: windows of stream 0, then of stream 1
    void* ws; size_t ws_bytes;
    unsigned* range_flag;
};
size_t pope_fine_preprocess_workspace(int M, int WW, int Cc, int Cf);
int pope_launch_fine_preprocess(const FinePreParams& q, hipStream_t stream);
int pope_launch_fine_match(const float* win0, const float* win1, int M, int Wn, int C, const float* mkpts1_c, float scale_px,
                           float* expec, float* mkpts1_f, hipStream_t stream);
size_t pope_loftr_layer_workspace(int n, int L, int S, int C, int H);
int pope_launch_loftr_layer(const LoftrLayerParams& p, hipStream_t stream);

// SAM image encoder (sam.hip; segment_anything/segment_anything/modeling/image_encoder.py:17-118).  Weight planes: scale
// K_PLANES_W_SCALE, torch Linear layout [out, in]; every pointer is a device pointer except `blocks`, `tap_blocks`, `tap_out`.
struct SamBlockParams {
    const float *norm1_w, *norm1_b;
    const void* qkv_wp; const float* qkv_b;      // [3 dim, dim], [3 dim]
    const void* proj_wp; const float* proj_b;    // [dim, dim], [dim]
    const float *rel_h, *rel_w;                  // gathered tables R[q][k][head_dim] (get_rel_pos, image_encoder.py:288-316): [s, s, hd]
                                                 // with s = window (window blocks) or grid (global blocks)
    const float *norm2_w, *norm2_b;
    const void* fc1_wp; const float* fc1_b;      // [hidden, dim]
    const void* fc2_wp; const float* fc2_b;      // [dim, hidden]
    int global;                                  // 1: global attention (window_size 0, image_encoder.py:77)
};
struct SamEncParams {
    const float* image;   // [B, 3, img, img] fp32
    float* out;           // [B, out_chans, g, g] fp32, g = img / patch
    int B, img, patch, dim, depth, heads, hidden, out_chans, window;
    int precision;        // POPE_PREC_F16X3 (weights = planes) | POPE_PREC_F16 (weights = f16 row-major, value * 256) |
                          // POPE_PREC_F32_MFMA (weights = fp32 matrices: the range guard's re-run, sam_f32.hip)
    float block_eps, neck_eps;   // LayerNorm eps of the blocks / of the neck's LayerNorm2d (<= 0: 1e-6)
    const void* patch_wp; const float* patch_b;   // [dim, 3 patch^2], [dim]
    const float* pos;                             // [g g, dim] or null (use_abs_pos = False)
    const float* ones;                            // [dim] of 1.0f (no LayerScale in this ViT)
    const SamBlockParams* blocks;                 // host array [depth]
    const void* neck0_wp;                         // [out_chans, dim]
    const float *neck1_w, *neck1_b;
    const void* neck2_wp;                         // [out_chans, 9 out_chans], taps (ky, kx, channel)
    const float *neck3_w, *neck3_b;
    int n_taps; const int* tap_blocks; float* const* tap_out;   // optional block outputs [B g g, dim] fp32
    void* ws; size_t ws_bytes;
    unsigned* range_flag;
};
size_t pope_sam_encoder_workspace(const SamEncParams& q);
int pope_launch_sam_encoder(const SamEncParams& q, hipStream_t stream);
// POPE_PREC_F32_MFMA twin (sam_f32.hip): every `*_wp` is a plain fp32 [out, in] matrix; same workspace; arguments validated by
// pope_launch_sam_encoder, which dispatches here
int pope_launch_sam_encoder_f32mfma(const SamEncParams& q, hipStream_t stream);

// Batched relative pose (pose.hip; src/utils/metrics.py:69-94): one workgroup per pair
struct PoseParams {
    const float* kpts0; const float* kpts1;   // [M, 2] fp32 pixel coordinates, the matches of pair b contiguous, pairs in order
    const int* counts;                        // [B] matches per pair (device; the matcher's counts)
    const double* K0; const double* K1;       // [B, 9] intrinsics, row-major
    int B; long long M;                       // M = rows of kpts0 / kpts1 (capacity; sum(counts) <= M)
    double thresh, conf;                      // pixels; RANSAC confidence
    int max_iters;
    unsigned long long seed;
    double* R; double* t; double* E;          // [B, 9], [B, 3], [B, 9]
    unsigned char* inliers;                   // [M]
    int* info;                                // [B, 8]: n_good (0 = None), RANSAC inliers, hypotheses, rounds, best hypothesis, best root, N, status
    void* xn; unsigned char* mask_ws; unsigned char* cheir_ws;   // filled in by the launcher from the workspace
    double* cand_ws; int* cmeta_ws;
};
size_t pope_pose_workspace(int B, long long M);
int pope_launch_estimate_pose(PoseParams q, void* ws, size_t ws_bytes, hipStream_t stream);
int pope_launch_five_point(const double* x0, const double* x1, int S, double* E_out, int* n_out, hipStream_t stream);
