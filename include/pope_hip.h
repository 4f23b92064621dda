/* pope_hip.h — C ABI of the MI355X-native (gfx950) POPE hot path.
 *
 * The reference (karltan0328/POPE) is pure Python on PyTorch: it has no FFI, its "operator
 * interface" for this path is the call convention of two nn.Modules plus a namespace
 * (SURVEY.md §8b).  Each entry point below names the reference call site it replaces; the
 * Python mirror in pope_amd/ binds them with ctypes (see INTEGRATION.md for the stub a
 * reference maintainer would add).
 *
 * Conventions: every pointer is a DEVICE pointer unless the name ends in _host; tensors are
 * dense row-major fp32 in the reference's native layouts (torch Linear weight = [out, in]);
 * `stream` is a hipStream_t; no entry point allocates or synchronises — workspaces are
 * caller-provided (size query functions); return value 0 = OK, negative = error
 * (pope_error_string).  All work is enqueued asynchronously.
 * Devices: work is launched on the device that owns `stream`; for stream == NULL (the default
 * stream, which does not name a device) on the calling thread's current device, so a caller with
 * several GPUs makes the operands' device current first (pope_amd/_lib.py:on_device_of does).
 * State: the library keeps, per device and filled lazily, the CU count and one "large dynamic
 * LDS enabled" bit per kernel (atomics; idempotent) — nothing else, and nothing per call, so
 * entry points may be called concurrently from several threads and for several devices.
 *
 * f16x3 range guard: POPE_PREC_F16X3 keeps operands as f16 pairs after a power-of-two scale
 * (activations x8, weights and matcher features x256), i.e. it needs |activation| < 8190 and
 * |weight| < 255.9.  Every entry point that converts values takes `range_flag`, an optional
 * DEVICE word (NULL = no check) into which POPE_RANGE_* bits are ORed when a converted value
 * would not be finite in f16 (or is not finite to begin with); the caller zeroes it, reads it at
 * its next synchronisation point and re-runs the work with POPE_PREC_F32_MFMA, which has no range
 * contract (pope_amd/dinov2.py, pope_amd/pipeline.py do exactly that).
 */
#ifndef POPE_HIP_H
#define POPE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POPE_ABI_VERSION 9

enum {
    POPE_EPI_BIAS = 0,        /* C = A.W^T + bias                         nn.Linear                     */
    POPE_EPI_BIAS_GELU = 1,   /* C = gelu_erf(A.W^T + bias)               mlp.py:35-41 (fc1 + nn.GELU)  */
    POPE_EPI_BIAS_LS_RES = 2  /* C = res + gamma*(A.W^T + bias)           block.py:105-106 + layer_scale.py:28 */
};

/* Arithmetic of the contractions.  Both take and return fp32 and accumulate in fp32:
 *  POPE_PREC_F32_MFMA  v_mfma_f32_32x32x2_f32, an exact k-ordered fp32 fma chain (157 TFLOP/s peak; on gfx950
 *                      this instruction runs on the VALU lanes);
 *  POPE_PREC_F16X3     operands split x = hi + lo (two f16, 22 significand bits), three f16 MFMAs
 *                      (v_mfma_f32_16x16x32_f16 in the GEMMs, v_mfma_f32_32x32x16_f16 in attention) per product
 *                      block on the matrix cores: same or smaller error than the fp32 chain inside the range
 *                      contract below (measured against fp64), 2.3x faster end to end. */
enum { POPE_PREC_F32_MFMA = 0, POPE_PREC_F16X3 = 1, POPE_PREC_F16 = 2 /* SAM encoder only: plain f16 operands, see there */ };
/* bits of a range_flag word: which f16x3 producer saw a value out of range */
enum { POPE_RANGE_PATCH = 1, POPE_RANGE_LAYERNORM = 2, POPE_RANGE_QKV = 4, POPE_RANGE_GELU = 8, POPE_RANGE_MATCH = 16,
       POPE_RANGE_INPUT = 32 };

int pope_abi_version(void);
const char* pope_error_string(int code);

/* ---- op-level entry points (each is one kernel; used by the parity tests) ------------- */

/* nn.LayerNorm(dim, eps) over the last dim — vision_transformer.py:90,230; block.py:56,68. */
int pope_layernorm_f32(const float* x, const float* weight, const float* bias, float* y,
                       int rows, int dim, float eps, void* stream);

/* nn.Linear (+ fused epilogue) — attention.py:51,60; mlp.py:35-44.  A[M,K], W[N,K], C[M,N];
 * gamma[N] and res[M,N] only for POPE_EPI_BIAS_LS_RES (res may alias C). */
int pope_linear_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                    int epilogue, const float* gamma, const float* res, void* stream);
/* Same with an explicit POPE_PREC_* (pope_linear_f32 == POPE_PREC_F32_MFMA).  POPE_PREC_F16X3 splits A and W inside
 * the kernel; with range_flag both operands are scanned first (POPE_RANGE_INPUT). */
int pope_linear_prec_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                         int epilogue, const float* gamma, const float* res, int precision,
                         unsigned* range_flag, void* stream);

/* f16x3 "planes": a tensor X[rows, cols] (cols % 32 == 0) kept as f16 halves with X * scale = hi + lo
 * (scale a power of two: POPE_PLANES_ACT_SCALE for activations, POPE_PLANES_W_SCALE for weights), laid
 * out row-major with 2*cols halves per row and, per 32-column chunk, the 32 hi halves followed by the
 * 32 lo halves (one 128-byte cache line per row and GEMM K-step):
 *     half_offset(row, col, plane) = row*2*cols + (col/32)*64 + plane*32 + col%32.
 * Producers split once (LayerNorm, the GELU epilogue, the weight loader) so that the f16x3 GEMM
 * stages MFMA-ready operands with no conversion work in its K loop. */
#define POPE_PLANES_ACT_SCALE 8.0f
#define POPE_PLANES_W_SCALE 256.0f
int pope_split_planes_f32(const float* src, void* planes, int rows, int cols, float scale,
                          unsigned* range_flag, void* stream);
/* nn.Linear on planes: A planes [M,K], W planes [N,K]; output either fp32 C[M,N] (c_planes == NULL) or
 * activation planes [M,N] (C == NULL; not for POPE_EPI_BIAS_LS_RES).  K % 32 == 0, K >= 64. */
int pope_linear_planes_f32(const void* a_planes, const void* w_planes, const float* bias, float* C,
                           void* c_planes, int M, int N, int K, int epilogue, const float* gamma,
                           const float* res, unsigned* range_flag, void* stream);

/* LayerNorm written as activation planes [rows, dim]. */
int pope_layernorm_planes_f32(const float* x, const float* weight, const float* bias, void* y_planes,
                              int rows, int dim, float eps, unsigned* range_flag, void* stream);

/* PatchEmbed.forward + prepare_tokens_with_masks — patch_embed.py:69-82,
 * vision_transformer.py:191-200.  img[B,3,H,W]; proj_w[dim, 3*patch*patch];
 * posb[ntok, dim] = {cls_token + pos[0]; conv_bias + pos[n]} with pos already interpolated to
 * the (H/patch, W/patch) grid; tokens[B, ntok, dim], ntok = 1 + (H/patch)*(W/patch). */
int pope_patch_embed_f32(const float* img, const float* proj_w, const float* posb, float* tokens,
                         int B, int H, int W, int patch, int dim, void* stream);
/* Same on the f16 matrix cores: the image patches are gathered into activation planes [B*ntok, kp] in
 * a_planes_scratch (>= B*ntok*kp*4 bytes; kp = 3*patch^2 rounded up to 32) and multiplied with proj_w_planes
 * [dim, kp] (weight planes, zero-padded columns); posb as above. */
int pope_patch_embed_planes_f32(const float* img, const void* proj_w_planes, const float* posb, float* tokens,
                                int B, int H, int W, int patch, int dim, void* a_planes_scratch,
                                size_t scratch_bytes, unsigned* range_flag, void* stream);

/* Attention.forward core — attention.py:51-59: qkv[B,N,3,heads,64] -> out[B,N,heads*64],
 * softmax((q*0.125) k^T) v. */
int pope_attention_f32(const float* qkv, float* out, int B, int N, int heads, void* stream);
/* Same with an explicit POPE_PREC_* (pope_attention_f32 == POPE_PREC_F32_MFMA); with range_flag a POPE_PREC_F16X3
 * call scans qkv first (POPE_RANGE_INPUT). */
int pope_attention_prec_f32(const float* qkv, float* out, int B, int N, int heads, int precision,
                            unsigned* range_flag, void* stream);
/* f16x3 attention on planes (layout and scale: POPE_PLANES_* above; the producer of the planes guards the range:
 * the output is a convex combination of v rows, so it fits whenever v did): qkv_planes [B*N, 3*heads*64] as written by
 * pope_linear_planes_f32(..., c_planes), out_planes [B*N, heads*64] as read by the proj GEMM.  heads*64 % 32 == 0. */
int pope_attention_planes_f32(const void* qkv_planes, void* out_planes, int B, int N, int heads, void* stream);

/* POPE_PREC_F16 attention core (BASELINE config 5's dtype; attention.py:51-59 in single-product f16 arithmetic with fp32 scores,
 * softmax and accumulators): qkv_f16 [B*N, 3*heads*64] f16 row-major holding q * (head_dim^-0.5 * log2 e), k, v — what the QKV
 * projection of the f16 ViT path writes — -> out_f16 [B*N, heads*64] f16, value * POPE_PLANES_ACT_SCALE (the f16 proj GEMM's
 * operand).  qkv_f16 and out_f16 16-byte aligned.  (ABI 9.) */
int pope_attention_f16(const void* qkv_f16, void* out_f16, int B, int N, int heads, void* stream);

/* Measurement only: pope_attention_planes_f32 through a diagnostic instantiation of the same kernel that counts the
 * (wave, 64-key tile) pairs whose softmax reference had to ADVANCE before the tile's exponentials (*exact_passes_host; of
 * B * heads * ceil(N / 32) * ceil(N / 64) pairs in all; until round 3 the count was of tiles redone after an overflow).
 * Same results; SYNCHRONISES the stream.  bench.py's `attention_ramp` leg. */
int pope_attention_planes_diag_f32(const void* qkv_planes, void* out_planes, int B, int N, int heads, long long* exact_passes_host,
                                   void* stream);

/* F.cosine_similarity(ref[1,D], fea[P,D], dim=1, eps) — eval_linemod_json.py:94. */
int pope_cls_cosine_f32(const float* ref, const float* fea, int P, int D, float eps, float* scores,
                        void* stream);

/* ---- whole-model entry point -------------------------------------------------------------- */

typedef struct pope_vit_block_weights {   /* state-dict keys blocks.{i}.*  (device pointers) */
    const float *norm1_w, *norm1_b;       /* norm1.weight/bias [dim]                          */
    const float *qkv_w, *qkv_b;           /* attn.qkv.weight [3dim,dim], bias [3dim]          */
    const float *proj_w, *proj_b;         /* attn.proj.weight [dim,dim], bias [dim]           */
    const float *ls1;                     /* ls1.gamma [dim]                                  */
    const float *norm2_w, *norm2_b;
    const float *fc1_w, *fc1_b;           /* mlp.fc1.weight [hidden,dim]                      */
    const float *fc2_w, *fc2_b;           /* mlp.fc2.weight [dim,hidden]                      */
    const float *ls2;
    /* optional (POPE_PREC_F16X3): weight planes [out][in] (layout above), scale POPE_PLANES_W_SCALE;
     * NULL -> the layer splits its fp32 weights on the fly */
    const void *qkv_wp, *fc1_wp, *fc2_wp, *proj_wp;
} pope_vit_block_weights;

typedef struct pope_vit_weights {
    int dim, depth, heads, patch, hidden;
    const float* patch_w;                 /* patch_embed.proj.weight flattened [dim, 3*patch^2] */
    const float *norm_w, *norm_b;         /* final norm                                         */
    const pope_vit_block_weights* blocks_host; /* HOST array [depth] of device-pointer structs  */
    int precision;                        /* POPE_PREC_* of the Linear layers and of attention; POPE_PREC_F16 (opt-in): the
                                             blocks' `*_wp` are plain f16 row-major matrices (value * 256), one MFMA per
                                             product; patch_wp stays weight planes (the patch embed is f16x3) */
    const void* patch_wp;                 /* optional: patch_w as weight planes [dim, kp], kp = 3*patch^2 rounded up to a
                                             multiple of 32 with zero columns; enables the f16x3 patch embed */
} pope_vit_weights;

size_t pope_vit_workspace_bytes(int B, int ntok, int dim, int hidden);

/* DinoVisionTransformer.forward_features — vision_transformer.py:221-236.
 * Outputs: x_prenorm[B,ntok,dim] (also the residual stream, required) and x_norm[B,ntok,dim]
 * (final LayerNorm; row 0 = x_norm_clstoken, rows 1.. = x_norm_patchtokens; may be NULL).
 * n_taps/tap_blocks_host/tap_out_host: optional copies of the residual stream after the listed
 * blocks (get_intermediate_layers, vision_transformer.py:238-288). */
int pope_vit_forward_f32(const pope_vit_weights* w_host, const float* img, int B, int H, int W,
                         const float* posb, float* x_prenorm, float* x_norm,
                         int n_taps, const int* tap_blocks_host, float* const* tap_out_host,
                         void* workspace, size_t workspace_bytes, unsigned* range_flag, void* stream);

/* In-situ kernel timing of the product path (bench.py's roofline leg): identical launches, plus
 * events_host[i] (hipEvent_t made by pope_event_create) recorded on `stream` immediately before
 * launch i and one closing event, so consecutive events bracket exactly one kernel.  kinds_host[i]
 * receives the POPE_K_* id of launch i; *n_launches_host the number of launches (= pope_vit_launch_count). */
enum { POPE_K_PATCH_EMBED = 0, POPE_K_LAYERNORM = 1, POPE_K_GEMM_QKV = 2, POPE_K_ATTENTION = 3,
       POPE_K_GEMM_PROJ = 4, POPE_K_GEMM_FC1 = 5, POPE_K_GEMM_FC2 = 6, POPE_K_TAP_COPY = 7 };
int pope_vit_launch_count(int depth);
/* Only launches whose POPE_K_* bit is set in kind_mask are bracketed (an event before and one after each; ~0u =
 * all): timing one kernel kind leaves every other launch of the sequence back to back, as in the untimed path.  events_host[i] / kinds_host[i] then hold the recorded events in order; kinds_host[i] is the kind
 * of the launch that STARTS at event i, or -1 for an event that only closes the previous bracket;
 * *n_launches_host = number of recorded events - 1. */
int pope_vit_forward_profiled_mask_f32(const pope_vit_weights* w_host, const float* img, int B, int H, int W,
                                       const float* posb, float* x_prenorm, float* x_norm,
                                       void* workspace, size_t workspace_bytes, unsigned* range_flag, void* stream,
                                       void* const* events_host, int n_events, int* kinds_host,
                                       int* n_launches_host, unsigned kind_mask);
int pope_event_create(void** event_host);
int pope_event_destroy(void* event);
int pope_event_elapsed_ms(void* start, void* stop, float* ms_host);  /* both events must have completed */

/* ---- dense matcher --------------------------------------------------------------------------- */

/* CoarseMatching.forward + get_coarse_match (eval, dual-softmax) —
 * src/matcher/utils/coarse_matching.py:106-119,151-261.
 * feat0[n,L,C], feat1[n,S,C] with `stride0`/`stride1` elements between consecutive pairs (L*C / S*C
 * when dense; larger when the features are the patch rows of an x_norm[B,1+L,C] buffer);
 * grids (h0,w0),(h1,w1); scale = hw0_i[0]/hw0_c[0].
 * conf_matrix[n,L,S]: the published confidence matrix (the drop-in CoarseMatching publishes it,
 * coarse_matching.py:145), or NULL for callers that only consume the match lists: the matrix then lives in the
 * workspace as sim only and is never written a second time.
 * Outputs have capacity n*L: b_ids,i_ids,j_ids (int64), mconf, mkpts0_c/mkpts1_c [.,2] (x,y);
 * counts[n+1] (int32): matches per pair, then the total M (read it after synchronising).
 * precision = POPE_PREC_* of the L x S x C contraction; POPE_PREC_F16X3 needs C % 32 == 0 (feat / sqrt(C) kept as
 * hi/lo planes, range-guarded: POPE_RANGE_MATCH); everything after the contraction is identical.
 * Workspace: pope_dense_match_workspace_bytes_prec(n, L, S, C, precision, conf_matrix != NULL). */
size_t pope_dense_match_workspace_bytes_prec(int n, int L, int S, int C, int precision, int publish_conf);
int pope_dense_match_prec_f32(const float* feat0, long long stride0, const float* feat1, long long stride1,
                              int n, int L, int S, int C, int h0, int w0, int h1, int w1,
                              float thr, int border_rm, float temperature, float scale,
                              float* conf_matrix, long long* b_ids, long long* i_ids, long long* j_ids,
                              float* mconf, float* mkpts0_c, float* mkpts1_c, int* counts,
                              void* workspace, size_t workspace_bytes, int precision, unsigned* range_flag,
                              void* stream);
/* Shorthands: POPE_PREC_F32_MFMA, conf_matrix published, no range flag. */
size_t pope_dense_match_workspace_bytes(int n, int L, int S);
int pope_dense_match_f32(const float* feat0, long long stride0, const float* feat1, long long stride1,
                         int n, int L, int S, int C,
                         int h0, int w0, int h1, int w1, float thr, int border_rm, float temperature,
                         float scale, float* conf_matrix, long long* b_ids, long long* i_ids,
                         long long* j_ids, float* mconf, float* mkpts0_c, float* mkpts1_c, int* counts,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- LoFTR encoder layer (first slice) ---------------------------------------------------- */

/* LoFTREncoderLayer.forward with LinearAttention — src/matcher/loftr_module/transformer.py:35-58,
 * linear_attention.py:20-47 (no masks): x[n,L,C] <- x + norm2(mlp(cat[x, norm1(merge(attention(q(x), k(source),
 * v(source))))])), in place; source[n,S,C] may be x itself ('self' layers).  C = 256 (coarse) or 128 (fine), nhead = 8.
 * The five bias-free Linears are given as f16x3 WEIGHT planes (layout above): q_wp [C,C], kv_wp [2C,C] (k_proj rows,
 * then v_proj rows), merge_wp [C,C], mlp0_wp [2C,2C], mlp1_wp [C,2C]; LayerNorm eps = ln_eps (nn.LayerNorm default
 * 1e-5).  LocalFeatureTransformer.forward (:85-106) is a sequence of these calls: 'self' = (f0,f0),(f1,f1); 'cross' =
 * (f0,f1) then (f1, NEW f0).
 * precision = POPE_PREC_F16X3, or POPE_PREC_F32_MFMA: the re-run of a range-guard event — the five `*_wp` are then plain fp32
 * [out, in] matrices and every contraction runs on the fp32 MFMA (no range contract; range_flag is not touched). */
typedef struct pope_loftr_layer_weights {
    const void *q_wp, *kv_wp, *merge_wp, *mlp0_wp, *mlp1_wp;
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} pope_loftr_layer_weights;
size_t pope_loftr_layer_workspace_bytes(int n, int L, int S, int C, int nhead);
int pope_loftr_encoder_layer_f32(const pope_loftr_layer_weights* w_host, float* x, const float* source,
                                 int n, int L, int S, int C, int nhead, float ln_eps, int precision,
                                 void* workspace, size_t workspace_bytes, unsigned* range_flag, void* stream);

/* ResNetFPN_8_2.forward — src/matcher/backbone/resnet_fpn.py:100-118 (BasicBlock :15-40), eval mode: the LoFTR
 * matcher's local-feature CNN.  gray[n,1,H,W] in [0,1], H and W multiples of 8.  Every convolution is an f16x3 planes
 * GEMM over zero-bordered NHWC activations (3x3 stride 1: implicit, no im2col; pope_amd/csrc/conv.hip); eval-mode
 * BatchNorm is folded into filter and bias by the caller.  Weights: planes (pope_split_planes_f32, scale 256) of
 * [Cout, K] matrices with K = taps x input channels rounded up to 32 (zero filled), tap-major (ky, kx, c):
 *   0 conv1+bn1 (K = 64: the 49 taps of the 7x7)            1..4  layer1: b0.conv1, b0.conv2, b1.conv1, b1.conv2
 *   5..9  layer2: b0.conv1, b0.conv2, b0.downsample (1x1), b1.conv1, b1.conv2        10..14 layer3: the same
 *   15 layer3_outconv  16 layer2_outconv  17 layer2_outconv2[0]+bn  18 layer2_outconv2[3]
 *   19 layer1_outconv  20 layer1_outconv2[0]+bn  21 layer1_outconv2[3]
 * b[i] = folded bias [Cout] or NULL (15, 16, 18, 19, 21).  Channel widths 128 / 196 / 256 (cvpr_ds_config.py).
 * Outputs are NHWC with a one-pixel border: out_c[n, H/8+2, W/8+2, 256] (x3_out; border zero) and
 * out_f[n, H/2+2, W/2+2, 128] (x1_out; border undefined) — the reference's NCHW maps are the interiors, permuted.
 * precision = POPE_PREC_F16X3, or POPE_PREC_F32_MFMA (the re-run of a range-guard event): w[i] are then the same [Cout, K]
 * matrices as plain fp32 and every convolution runs on the fp32 MFMA (gemm_f32.hip, implicit 3x3 loader). */
typedef struct pope_resnetfpn_weights {
    const void* w[22];
    const float* b[22];
} pope_resnetfpn_weights;
size_t pope_resnetfpn_workspace_bytes(int n, int H, int W);
int pope_resnetfpn_forward_f32(const pope_resnetfpn_weights* w_host, const float* gray, int n, int H, int W, int precision,
                               float* out_c, float* out_f, void* workspace, size_t workspace_bytes,
                               unsigned* range_flag, void* stream);

/* FinePreprocess.forward — src/matcher/loftr_module/fine_preprocess.py:29-59 (fine_concat_coarse_feat = True), for
 * M > 0 matches: Wn x Wn windows (zero padded by Wn/2) of the 1/2-resolution maps centred on the matched coarse
 * cells (i_ids in map 0, j_ids in map 1; cell id = cy * wc + cx, centre pixel = cell * stride), concatenated with
 * down_proj(coarse feature of the cell) and passed through merge_feat.  Only the M matched windows are gathered (the
 * reference unfolds all of them first).  feat_f*: fp32 maps addressed through ELEMENT strides strides*_host[4] =
 * (n, c, h, w), so NCHW tensors and NHWC views both work; feat_c0[n,L,Cc], feat_c1[n,S,Cc]; b/i/j_ids: int64[M];
 * down_wp[Cf,Cc], merge_wp[Cf,2*Cf]: weight planes (scale 256) — plain fp32 matrices for precision = POPE_PREC_F32_MFMA, the
 * re-run of a range-guard event —, biases fp32; out[2*M, Wn*Wn, Cf] = windows of stream 0 then stream 1 (the reference's
 * torch.chunk(…, 2)). */
size_t pope_fine_preprocess_workspace_bytes(int M, int Wn, int Cc, int Cf);
int pope_fine_preprocess_f32(const float* feat_f0, const long long* strides0_host, int H0, int W0, int wc0,
                             const float* feat_f1, const long long* strides1_host, int H1, int W1, int wc1,
                             const float* feat_c0, const float* feat_c1, int L, int S, int Cc, int Cf,
                             const long long* b_ids, const long long* i_ids, const long long* j_ids, int M,
                             int Wn, int stride, const void* down_wp, const float* down_b, const void* merge_wp,
                             const float* merge_b, int precision, float* out, void* workspace, size_t workspace_bytes,
                             unsigned* range_flag, void* stream);
/* FineMatching.forward + get_fine_match — src/matcher/utils/fine_matching.py:15-74 for M > 0: correlation of the
 * centre of window 0 with window 1 (temperature 1/sqrt(C)), softmax, expectation over linspace(-1,1,Wn)^2 as (x, y)
 * and the summed standard deviation -> expec_f[M,3]; mkpts1_f[M,2] = mkpts1_c + expec_xy * (Wn/2) * scale_px
 * (scale_px = hw0_i[0] / hw0_f[0]).  win0, win1: [M, Wn*Wn, C] fp32 (the fine transformer's outputs). */
int pope_fine_match_f32(const float* win0, const float* win1, int M, int Wn, int C, const float* mkpts1_c,
                        float scale_px, float* expec_f, float* mkpts1_f, void* stream);

/* ---- SAM image encoder (BASELINE config 5) --------------------------------------------------- */

/* ImageEncoderViT.forward — segment_anything/segment_anything/modeling/image_encoder.py:107-118 as build_sam.py:66-79
 * configures it (LayerNorm eps 1e-6, qkv bias, absolute + decomposed relative position terms, window attention with
 * `window` x `window` windows except in the blocks marked global, MLP ratio = hidden / dim, neck 1x1 conv ->
 * LayerNorm2d -> 3x3 conv -> LayerNorm2d).  image[B,3,img,img] fp32 (already normalised and padded by the caller,
 * sam.py preprocess) -> out[B,out_chans,g,g] fp32, g = img / patch.  head_dim = dim / heads must be 64 (ViT-B/L) or
 * 80 (ViT-H); dim % 128 == 0; out_chans % 256 == 0; patch % 8 == 0.  precision POPE_PREC_F16X3: fp32 operands as
 * hi + lo f16 planes, three MFMAs per product; POPE_PREC_F16: plain f16 operands, one MFMA per product (see the field);
 * fp32 accumulation, softmax, LayerNorm, GELU and residual stream in both.
 * Weights: `*_wp` = weight planes (pope_split_planes_f32, scale 256; POPE_PREC_F16: f16 row-major, value * 256) of the
 * torch [out, in] matrices —
 * patch_wp[dim, 3 patch^2] = proj.weight.reshape(dim, -1); neck0_wp[out_chans, dim]; neck2_wp[out_chans, 9 out_chans]
 * with the taps in (ky, kx, channel) order = weight.permute(0, 2, 3, 1).reshape(out_chans, -1).  pos[g*g, dim] or NULL.
 * rel_h / rel_w: the tables get_rel_pos returns (image_encoder.py:288-316), R[q][k][head_dim] fp32 with q, k < window
 * (window blocks) or < g (global blocks); the bias q.Rh + q.Rw is folded into the score product (sam.hip).
 * ones[dim] = 1.0f.  blocks_host: HOST array [depth] of device pointers.  Optional taps as pope_vit_forward_f32
 * (tap_out_host[t][B*g*g, dim] fp32 = output of block tap_blocks_host[t]). */
typedef struct pope_sam_block_weights {
    const float *norm1_w, *norm1_b;
    const void* qkv_wp; const float* qkv_b;
    const void* proj_wp; const float* proj_b;
    const float *rel_h, *rel_w;
    const float *norm2_w, *norm2_b;
    const void* fc1_wp; const float* fc1_b;
    const void* fc2_wp; const float* fc2_b;
    int global_attn;
} pope_sam_block_weights;
typedef struct pope_sam_encoder_weights {
    int img, patch, dim, depth, heads, hidden, out_chans, window;
    int precision;   /* POPE_PREC_F16X3, or POPE_PREC_F16: BASELINE config 5's "fp16" — every `*_wp` is then a plain f16
                      * row-major matrix (value * 256) and every contraction ONE f16 MFMA per product with fp32
                      * accumulation; residual stream, softmax, LayerNorm statistics and GELU stay fp32 —, or
                      * POPE_PREC_F32_MFMA: every `*_wp` is a plain fp32 matrix and every contraction runs on the fp32 MFMA (the
                      * reference's arithmetic, no range contract: what a range-guard event is re-run in; ~10x slower) */
    const void* patch_wp; const float* patch_b;
    const float* pos;
    const float* ones;
    const pope_sam_block_weights* blocks_host;
    const void* neck0_wp;
    const float *neck1_w, *neck1_b;
    const void* neck2_wp;
    const float *neck3_w, *neck3_b;
    float block_eps, neck_eps;   /* LayerNorm eps of the blocks' norm1 / norm2 (build_sam.py:71: 1e-6; nn.LayerNorm's default: 1e-5)
                                  * and of the neck's two LayerNorm2d (common.py:28: 1e-6); <= 0 selects 1e-6 */
} pope_sam_encoder_weights;
size_t pope_sam_encoder_workspace_bytes(const pope_sam_encoder_weights* w_host, int B);
int pope_sam_encoder_forward_f32(const pope_sam_encoder_weights* w_host, const float* image, int B, float* out,
                                 int n_taps, const int* tap_blocks_host, float* const* tap_out_host,
                                 void* workspace, size_t workspace_bytes, unsigned* range_flag, void* stream);

/* ---- caller-side preprocessing, batched ------------------------------------------------- */

/* set_torch_image for P crops at once — segment_anything/segment_anything/dinov2_utils.py:55-78: Resize (Pillow's
 * 8-bit bilinear resample, bit-identical) -> CenterCrop -> ToTensor -> Normalize.  img_hwc[P,Hin,Win,3] uint8 (channel
 * order as given: the drivers pass BGR); the window / weight tables of the two resample passes are DEVICE int32 arrays
 * built by the host exactly as Pillow builds them (pope_amd/preprocess.py:resize_tables): hstart/hcount[OW],
 * hk[OW,kh], vstart/vcount[OH], vk[OH,kv] with 22 fractional bits; (top,left,ch,cw) = crop window in the resized
 * image; [row0, row0+nrows) = input rows those output rows read; mean_host/std_host[3] = HOST floats;
 * out[P,3,ch,cw] fp32; scratch >= P*nrows*cw*3 bytes. */
int pope_preprocess_u8_f32(const unsigned char* img_hwc, int P, int Hin, int Win,
                           const int* hstart, const int* hcount, const int* hk, int kh,
                           const int* vstart, const int* vcount, const int* vk, int kv,
                           int top, int left, int ch, int cw, int row0, int nrows,
                           const float* mean_host, const float* std_host, float* out,
                           unsigned char* scratch, size_t scratch_bytes, void* stream);
/* ToTensor + Normalize of a crop window without resizing (torchvision semantics: x / 255, then (x - mean) / std in
 * fp32) — the dense pair path's 476 x 630 centre crop of a 640 x 480 frame: img_hwc[P,Hin,Win,3] uint8 ->
 * out[P,3,ch,cw] fp32 (16-byte aligned); mean_host / std_host[3] = HOST floats in the channel order of img. */
int pope_crop_normalize_u8_f32(const unsigned char* img_hwc, int P, int Hin, int Win, int top, int left, int ch, int cw,
                               const float* mean_host, const float* std_host, float* out, void* stream);
/* cv2.cvtColor(BGR2GRAY) (8-bit fixed point) followed by / 255. — eval_linemod_json.py:103-111:
 * bgr_hwc[P,H,W,3] uint8 -> out[P,1,H,W] fp32 in [0,1] (the Matcher's input). */
int pope_gray_u8_f32(const unsigned char* bgr_hwc, int P, int H, int W, float* out, void* stream);

/* Proposal crops — get_image_crop_resize (utils/data_utils.py:239-255 = cv2.warpAffine(image, M, (w, h), INTER_LINEAR), border
 * constant 0) as the drivers use it twice per SAM proposal (eval_linemod_json.py:83-90: crop at the expanded box's own size,
 * an integer translation, then a uniform resize of that crop to 256 x 256), for P proposals of one frame in one launch.
 * img_hwc[H, W, C] uint8 (C <= 4); minv[P, 6] fp64 (DEVICE): the INVERSE 2 x 3 map of each output (destination pixel ->
 * source position in window coordinates; what cv::warpAffine derives from the forward matrix); win[P, 4] int32 (DEVICE) =
 * (x0, y0, w, h): the source of output p is the w x h window of the frame at (x0, y0), everything outside the window or the
 * frame reads 0 — i.e. the reference's intermediate zero-padded crop, never materialised; (0, 0, W, H) warps the frame itself.
 * out[P, oh, ow, C] uint8.  Arithmetic: OpenCV's 8-bit bilinear convention (1/32 px positions, integer weights summing to
 * 1024, round to nearest) — restated, unpinned against cv2 (absent); integer translations are exact copies. */
int pope_crop_warp_u8(const unsigned char* img_hwc, int H, int W, int C, const double* minv, const int* win, int P,
                      int oh, int ow, unsigned char* out, void* stream);

/* ---- relative pose from the matches ----------------------------------------------------------- */

/* estimate_pose for B pairs in one launch — src/utils/metrics.py:69-94 (call site eval_linemod_json.py:160): K-normalise
 * both point sets (:72-75), threshold = thresh / mean(fx0, fy1, fx0, fy1) (:78), essential matrix by RANSAC over five-point
 * minimal samples (what cv2.findEssentialMat(..., cv2.RANSAC) does: Sampson error <= threshold^2, a model is kept when it has
 * MORE inliers than the best so far and at least five, budget log(1 - conf) / log(1 - w^5) re-evaluated after every round of
 * 256 hypotheses, at most max_iters <= 2^27 — OpenCV's default is 1000), then recoverPose for every returned E (:86-94): the
 * (R, t) of the four decompositions with the most inliers in front of both cameras.  A pair with exactly five matches is
 * the minimal problem itself: all of its solutions go through recoverPose.  All arithmetic fp64.
 * Inputs are the dense matcher's compacted outputs as they lie in HBM: kpts0 / kpts1 [M, 2] fp32 pixel coordinates with
 * the matches of pair b contiguous and pairs in order (mkpts0_c / mkpts1_c or mkpts0_f / mkpts1_f), counts [B] int32
 * (DEVICE: matches per pair, sum <= M), K0 / K1 [B, 9] fp64 row-major intrinsics of image 0 / image 1 of each pair.
 * Outputs: R [B, 9], t [B, 3] (unit norm), E [B, 9] fp64; inliers [M] bytes (RANSAC inlier AND in front of both cameras,
 * the mask recoverPose leaves behind); info [B, 8] int32 = {n_inliers of the returned pose — 0 means `None` (fewer than five
 * matches, no model with five inliers, or no point in front of both cameras) —, RANSAC inliers, hypotheses tried, rounds,
 * winning hypothesis, winning root, matches of the pair, status (-1: refused — the counts up to and including this pair exceed
 * M, or one of them is negative)}.  Rows of `inliers` past sum(counts) are not written.
 * The minimal samples of hypothesis h come from a counter-based hash of (seed, h): results do not depend on the batch a
 * pair rides in.  cv2's own random stream is not reproducible without cv2 (absent here): parity with OpenCV is unpinned,
 * the checker is oracle/pose_ref.py (same algorithm, same samples, numpy fp64).
 * Workspace (>= pope_estimate_pose_workspace_bytes(B, M), 32-byte aligned): normalised points and two masks per match, and per
 * pair the list of one round's models (256 hypotheses x <= 10 roots x 9 fp64) that spreads the scoring evenly over the threads. */
size_t pope_estimate_pose_workspace_bytes(int B, long long M);
int pope_estimate_pose_f64(const float* kpts0, const float* kpts1, const int* counts, const double* K0, const double* K1,
                           int B, long long M, double thresh, double conf, int max_iters, unsigned long long seed,
                           double* R, double* t, double* E, unsigned char* inliers, int* info,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The minimal solver alone (parity tests): S problems of five correspondences in normalised coordinates,
 * x0 / x1 [S, 5, 2] fp64 -> E_out [S, 10, 9] (unit Frobenius norm, ascending root order, zero filled), n_out [S]. */
int pope_five_point_f64(const double* x0, const double* x1, int S, double* E_out, int* n_out, void* stream);

/* ---- host-side helper ------------------------------------------------------------------------ */

/* Streaming top-3 proposal vote — eval_linemod_json.py:71,95-101 (HOST pointers): slots start at
 * 0; a proposal enters iff score > min(slots), replacing the first minimal slot. */
int pope_streaming_top3_host(const float* scores_host, int P, float* slot_scores_host,
                             long long* slot_index_host);

#ifdef __cplusplus
}
#endif
#endif /* POPE_HIP_H */
