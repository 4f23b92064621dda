// C ABI (include/pope_hip.h) over the kernel launchers: argument checks, workspace carving and the
// launch sequence of one DinoVisionTransformer forward.  No allocation, no synchronisation.
#include "../../include/pope_hip.h"
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Every launching entry point runs on the device that owns `stream` (the reference keeps the matcher on cuda:1 while
// cuda:0 is current, pope_model_api.py:181-184): the launchers' per-device state (LDS opt-in, CU count) and the
// launches themselves then belong to the right GPU whatever the caller's current device is.  NULL = the current
// device's default stream.  The previous device is restored on return.
struct StreamDevice {
    int prev = -1;
    bool switched = false;
    explicit StreamDevice(void* stream) {
        int dev = -1;
        if (!stream || hipGetDevice(&prev) != hipSuccess) return;
        if (hipStreamGetDevice(static_cast<hipStream_t>(stream), &dev) == hipSuccess && dev != prev)
            switched = hipSetDevice(dev) == hipSuccess;
    }
    ~StreamDevice() {
        if (switched) (void)hipSetDevice(prev);
    }
    StreamDevice(const StreamDevice&) = delete;
    StreamDevice& operator=(const StreamDevice&) = delete;
};

// Optional in-situ timing: events[i] is recorded on the stream right before launch i and one more
// after the last launch, so events[i]..events[i+1] bracket exactly one kernel of the product path.
// With a kind mask only the selected launches are bracketed: an event is recorded when the coming launch is
// selected (it starts a bracket) or the previous one was (it closes one); kinds[i] = -1 marks close-only events.
struct Recorder {
    void* const* events;
    int capacity;
    int* kinds;
    int n;
    unsigned mask = ~0u;
    bool open = false;
    bool mark(int kind, hipStream_t stream) {
        if (!events) return true;
        const bool sel = kind >= 0 && ((mask >> kind) & 1u);
        if (!sel && !open) return true;
        if (n >= capacity) return false;
        if (hipEventRecord(static_cast<hipEvent_t>(events[n]), stream) != hipSuccess) return false;
        if (kinds) kinds[n] = sel ? kind : -1;
        open = sel;
        ++n;
        return true;
    }
};

__global__ __launch_bounds__(256) void cls_cosine_kernel(const float* __restrict__ ref, const float* __restrict__ fea,
                                                          int P, int D, float eps, float* __restrict__ scores) {
    // x.y / (max(|x|, eps) * max(|y|, eps)) — torch semantics, each norm clamped separately
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const float* f = fea + size_t(p) * D;
    float dot = 0.f, nr = 0.f, nf = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float a = ref[i], b = f[i];
        dot += a * b;
        nr += a * a;
        nf += b * b;
    }
    dot = wave_sum(dot);
    nr = wave_sum(nr);
    nf = wave_sum(nf);
    if (lane == 0) scores[p] = dot / (fmaxf(sqrtf(nr), eps) * fmaxf(sqrtf(nf), eps));
}

}  // namespace

extern "C" {

int pope_abi_version(void) { return POPE_ABI_VERSION; }

const char* pope_error_string(int code) {
    switch (code) {
        case POPE_OK: return "ok";
        case POPE_ERR_ARG: return "invalid argument (shape, alignment or null pointer)";
        case POPE_ERR_LAUNCH: return "HIP launch failed";
        case POPE_ERR_WORKSPACE: return "workspace too small";
    }
    return "unknown error";
}

int pope_layernorm_f32(const float* x, const float* weight, const float* bias, float* y, int rows, int dim,
                       float eps, void* stream) {
    StreamDevice on_device(stream);
    if (!x || !weight || !bias || !y) return POPE_ERR_ARG;
    return pope_launch_layernorm_f32(x, dim, weight, bias, y, dim, rows, dim, eps, static_cast<hipStream_t>(stream));
}

int pope_linear_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                    int epilogue, const float* gamma, const float* res, void* stream) {
    return pope_linear_prec_f32(A, W, bias, C, M, N, K, epilogue, gamma, res, POPE_PREC_F32_MFMA, nullptr, stream);
}

int pope_linear_prec_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                         int epilogue, const float* gamma, const float* res, int precision, unsigned* range_flag,
                         void* stream) {
    StreamDevice on_device(stream);
    if (!A || !W || !C || epilogue < 0 || epilogue > POPE_EPI_BIAS_LS_RES) return POPE_ERR_ARG;
    if (precision != POPE_PREC_F32_MFMA && precision != POPE_PREC_F16X3) return POPE_ERR_ARG;
    GemmParams g = {};
    g.A = A; g.W = W; g.bias = bias; g.C = C;
    g.lda = K; g.ldw = K; g.ldc = N;
    g.M = M; g.N = N; g.K = K;
    g.epilogue = epilogue;
    g.gamma = gamma; g.res = res; g.ldres = N;
    g.range_flag = range_flag;
    // shapes the f16x3 kernel does not take (K % 32 != 0) run on the fp32 MFMA: same contract, same results
    if (precision == POPE_PREC_F16X3 && pope_gemm_f16x3_supported(g))
        return pope_launch_gemm_nt_f16x3(g, static_cast<hipStream_t>(stream));
    return pope_launch_gemm_nt_f32(g, static_cast<hipStream_t>(stream));
}

int pope_split_planes_f32(const float* src, void* planes, int rows, int cols, float scale, unsigned* range_flag, void* stream) {
    StreamDevice on_device(stream);
    return pope_launch_split_planes(src, planes, rows, cols, scale, range_flag, static_cast<hipStream_t>(stream));
}

int pope_linear_planes_f32(const void* a_planes, const void* w_planes, const float* bias, float* C, void* c_planes,
                           int M, int N, int K, int epilogue, const float* gamma, const float* res, unsigned* range_flag,
                           void* stream) {
    StreamDevice on_device(stream);
    if (epilogue < 0 || epilogue > POPE_EPI_BIAS_LS_RES) return POPE_ERR_ARG;
    GemmParams g = {};
    g.range_flag = range_flag;
    g.range_bit = epilogue == POPE_EPI_BIAS_GELU ? POPE_RANGE_GELU : POPE_RANGE_QKV;
    g.a_pl = a_planes; g.w_pl = w_planes;
    g.bias = bias; g.C = C; g.c_pl = c_planes;
    g.lda = K; g.ldw = K; g.ldc = N;
    g.M = M; g.N = N; g.K = K;
    g.epilogue = epilogue;
    g.gamma = gamma; g.res = res; g.ldres = N;
    return pope_launch_gemm_nt_f16x3_planes(g, static_cast<hipStream_t>(stream));
}

int pope_layernorm_planes_f32(const float* x, const float* weight, const float* bias, void* y_planes, int rows, int dim,
                              float eps, unsigned* range_flag, void* stream) {
    StreamDevice on_device(stream);
    if (!x || !weight || !bias) return POPE_ERR_ARG;
    return pope_launch_layernorm_planes(x, dim, weight, bias, y_planes, rows, dim, eps, range_flag, static_cast<hipStream_t>(stream));
}

int pope_patch_embed_f32(const float* img, const float* proj_w, const float* posb, float* tokens, int B, int H,
                         int W, int patch, int dim, void* stream) {
    StreamDevice on_device(stream);
    if (!img || !proj_w || !posb || !tokens || B <= 0 || patch <= 0 || H % patch || W % patch) return POPE_ERR_ARG;
    GemmParams g = {};
    g.A = img; g.W = proj_w; g.C = tokens;
    g.K = 3 * patch * patch;
    g.ldw = g.K; g.ldc = dim;
    g.ntok = 1 + (H / patch) * (W / patch);
    g.M = B * g.ntok; g.N = dim;
    g.epilogue = EPI_POSB;
    g.posb = posb;
    g.img_h = H; g.img_w = W; g.patch = patch; g.grid_w = W / patch;
    return pope_launch_gemm_nt_f32(g, static_cast<hipStream_t>(stream));
}

int pope_patch_embed_planes_f32(const float* img, const void* proj_w_planes, const float* posb, float* tokens, int B, int H,
                                int W, int patch, int dim, void* a_planes_scratch, size_t scratch_bytes, unsigned* range_flag,
                                void* stream_) {
    StreamDevice on_device(stream_);
    if (!img || !proj_w_planes || !posb || !tokens || !a_planes_scratch || B <= 0 || patch <= 0 || H % patch || W % patch)
        return POPE_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int kp = (3 * patch * patch + 31) & ~31, ntok = 1 + (H / patch) * (W / patch);
    if (scratch_bytes < size_t(B) * ntok * kp * 4) return POPE_ERR_WORKSPACE;
    int rc = pope_launch_im2col_planes(img, a_planes_scratch, B, H, W, patch, kp, range_flag, stream);
    if (rc) return rc;
    // tokens[b, n] = posb[n] + 1 * (A[b, n] . W^T): rows n = 0 are all-zero A rows (cls_token + pos_embed[0] from the table)
    GemmParams g = {};
    g.a_pl = a_planes_scratch; g.w_pl = proj_w_planes;
    g.C = tokens;
    g.lda = kp; g.ldw = kp; g.ldc = dim;
    g.M = B * ntok; g.N = dim; g.K = kp;
    g.epilogue = EPI_BIAS_LS_RES;
    g.res = posb; g.ldres = dim; g.res_mod = ntok;
    return pope_launch_gemm_nt_f16x3_planes(g, stream);
}

int pope_attention_planes_f32(const void* qkv_planes, void* out_planes, int B, int N, int heads, void* stream) {
    StreamDevice on_device(stream);
    if (!qkv_planes || !out_planes) return POPE_ERR_ARG;
    return pope_launch_attention_f16x3_planes_io(qkv_planes, out_planes, B, N, heads, static_cast<hipStream_t>(stream));
}

int pope_attention_f16(const void* qkv_f16, void* out_f16, int B, int N, int heads, void* stream) {
    StreamDevice on_device(stream);
    return pope_launch_attention_f16_dma(qkv_f16, out_f16, B, N, heads, static_cast<hipStream_t>(stream));
}

int pope_attention_planes_diag_f32(const void* qkv_planes, void* out_planes, int B, int N, int heads, long long* exact_passes_host,
                                   void* stream) {
    StreamDevice on_device(stream);
    if (!qkv_planes || !out_planes) return POPE_ERR_ARG;
    return pope_launch_attention_f16x3_planes_io_diag(qkv_planes, out_planes, B, N, heads, exact_passes_host, static_cast<hipStream_t>(stream));
}

int pope_attention_f32(const float* qkv, float* out, int B, int N, int heads, void* stream) {
    return pope_attention_prec_f32(qkv, out, B, N, heads, POPE_PREC_F32_MFMA, nullptr, stream);
}

int pope_attention_prec_f32(const float* qkv, float* out, int B, int N, int heads, int precision, unsigned* range_flag,
                            void* stream) {
    StreamDevice on_device(stream);
    if (!qkv || !out) return POPE_ERR_ARG;
    if (precision == POPE_PREC_F16X3 && range_flag) {  // q, k, v are split inside the kernel: check them in a scan
        if (B <= 0 || N <= 0 || heads <= 0) return POPE_ERR_ARG;
        const int rc = pope_launch_range_check(qkv, size_t(B) * N * 3 * heads * 64, 1.0f, range_flag, POPE_RANGE_INPUT,
                                               static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    if (precision == POPE_PREC_F16X3) return pope_launch_attention_f16x3(qkv, out, B, N, heads, static_cast<hipStream_t>(stream));
    if (precision != POPE_PREC_F32_MFMA) return POPE_ERR_ARG;
    return pope_launch_attention_f32(qkv, out, B, N, heads, static_cast<hipStream_t>(stream));
}

int pope_cls_cosine_f32(const float* ref, const float* fea, int P, int D, float eps, float* scores, void* stream) {
    StreamDevice on_device(stream);
    if (!ref || !fea || !scores || P <= 0 || D <= 0) return POPE_ERR_ARG;
    hipLaunchKernelGGL(cls_cosine_kernel, dim3((P + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), ref, fea,
                       P, D, eps, scores);
    return pope_check_launch();
}

size_t pope_vit_workspace_bytes(int B, int ntok, int dim, int hidden) {
    if (B <= 0 || ntok <= 0 || dim <= 0 || hidden <= 0) return 0;
    const size_t rows = size_t(B) * ntok;
    const size_t big = size_t(hidden) > size_t(4) * dim ? size_t(hidden) : size_t(4) * dim;
    return align_up(rows * dim * sizeof(float), 256) + align_up(rows * big * sizeof(float), 256);
}

static int vit_forward_impl(const pope_vit_weights* w, const float* img, int B, int H, int W, const float* posb,
                            float* x_prenorm, float* x_norm, int n_taps, const int* tap_blocks_host,
                            float* const* tap_out_host, void* workspace, size_t workspace_bytes, unsigned* range_flag,
                            void* stream_, Recorder& rec) {
    StreamDevice on_device(stream_);
    if (!w || !img || !posb || !x_prenorm || !workspace || !w->blocks_host) return POPE_ERR_ARG;
    if (w->dim != w->heads * 64 || w->patch <= 0 || H % w->patch || W % w->patch || B <= 0) return POPE_ERR_ARG;
    if (n_taps < 0 || (n_taps > 0 && (!tap_blocks_host || !tap_out_host))) return POPE_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int dim = w->dim, hidden = w->hidden, prec = w->precision;
    if (prec != POPE_PREC_F32_MFMA && prec != POPE_PREC_F16X3 && prec != POPE_PREC_F16) return POPE_ERR_ARG;
    const int ntok = 1 + (H / w->patch) * (W / w->patch);
    const int rows = B * ntok;
    if (workspace_bytes < pope_vit_workspace_bytes(B, ntok, dim, hidden)) return POPE_ERR_WORKSPACE;

    // workspace: xn [rows,dim] | big [rows, max(4dim, hidden)] = {qkv [rows,3dim], attn [rows,dim]} or fc1 out
    char* ws = static_cast<char*>(workspace);
    float* xn = reinterpret_cast<float*>(ws);
    float* big = reinterpret_cast<float*>(ws + align_up(size_t(rows) * dim * sizeof(float), 256));
    float* qkv = big;
    float* att = big + size_t(rows) * 3 * dim;
    float* hid = big;
    float* x = x_prenorm;
    const float eps = 1e-6f;  // vision_transformer.py:90

    // f16x3 = the planes dataflow end to end (every operand is split ONCE by its producer, which also guards the f16
    // range: range_flag).  It needs the weight planes of all four Linear layers of every block; without them (or with
    // a width the planes layout does not take) the model runs on the fp32 MFMA, which has no range contract.
    bool planes = (prec == POPE_PREC_F16X3 || prec == POPE_PREC_F16) && dim % 32 == 0 && dim >= 64 && hidden % 32 == 0;
    for (int i = 0; planes && i < w->depth; ++i) {
        const pope_vit_block_weights& k = w->blocks_host[i];
        planes = k.qkv_wp && k.proj_wp && k.fc1_wp && k.fc2_wp;
    }
    // POPE_PREC_F16: the blocks' Linear layers and attention in plain f16 (one MFMA per product; `*_wp` of the blocks
    // are f16 row-major matrices, value * 256); the patch embed stays f16x3 (`patch_wp` = planes) and the residual
    // stream, LayerNorm statistics, softmax and GELU fp32.  Needs the weights and widths the plain GEMM takes.
    const bool plain = prec == POPE_PREC_F16;
    if (plain && (!planes || !w->patch_wp || (dim & 63) || (hidden & 63))) return POPE_ERR_ARG;
    const int f32 = POPE_PREC_F32_MFMA;

#define POPE_MARK(kind) do { if (!rec.mark(kind, stream)) return POPE_ERR_ARG; } while (0)
#define POPE_TRY(call) do { if ((rc = (call))) return rc; } while (0)
    int rc;
    void* xn_pl = xn;    // planes alias the xn / fc1 buffers: 2 x f16 per element = the fp32 footprint
    void* hid_pl = hid;
    // Fused form (dim 384): every residual GEMM (patch embed, proj, fc2) also emits the LayerNorm that follows it —
    // as planes for the next GEMM, or as fp32 x_norm after the last block — so no stand-alone LayerNorm launch is left
    // (gemm_rowln.hip).
    GemmParams probe = {};
    probe.M = rows; probe.N = dim; probe.K = dim; probe.lda = dim; probe.ldw = dim; probe.ldc = dim; probe.ldres = dim;
    const bool fusable = planes && !plain && w->patch_wp && pope_gemm_rowln_supported(probe);
    // Small batches: the full-row-tile kernel has one tile per 128 rows, each a serial chain of K / 32 K-steps + a 23 us
    // epilogue; while the 128 x 128 residual GEMM still fits ONE round of its 2 x CUs workgroup slots (3 column tiles per row
    // tile) it finishes sooner, and `layernorm_rowln_order` reproduces the fused epilogue's LayerNorm bit for bit — an image
    // gives the same tokens alone (this path) and inside a 64-image chunk (fused path).  Driver step (9 images of 196 x
    // 196): proj 38 -> ~25 us, FC2 105 -> ~70 us per launch.
    const bool small = fusable && 3 * ((rows + 127) / 128) <= 2 * pope_cu_count();
    const bool fused = fusable && !small;
    // plain GEMM over the token rows: C (fp32) or c_f16 (f16 row-major) = epi(a_f16 . w_f16^T + bias [...])
    auto plain_gemm = [&](const void* a_f16, const void* w_f16, const float* bias, float* Cf, void* c_f16, int N, int K, int epi,
                          const float* gamma, const float* res) {
        GemmParams g = {};
        g.range_flag = range_flag;
        g.range_bit = epi == EPI_BIAS_GELU ? POPE_RANGE_GELU : POPE_RANGE_QKV;
        g.a_pl = a_f16; g.w_pl = w_f16; g.bias = bias; g.C = Cf; g.c_pl = c_f16;
        g.lda = K / 2; g.ldw = K / 2; g.K = K / 2; g.ldc = c_f16 ? N / 2 : N;   // column pairs (GemmParams::plain)
        g.M = rows; g.N = N; g.epilogue = epi; g.gamma = gamma; g.res = res; g.ldres = N;
        g.plain = 1;
        if (epi == EPI_QKV_F16) { g.sam_dim = N / 3; g.sam_qscale = 0.125f * 1.44269504088896340736f; }   // heads of 64: head_dim^-0.5 * log2 e
        return pope_launch_gemm_nt_f16x3_planes(g, stream);
    };
    // residual GEMM + following LayerNorm: x = res + gamma * (a . W^T + bias); LN(x; ln_w, ln_b) -> planes or fp32
    auto rowln = [&](const void* a_pl, const void* w_pl, int K, const float* bias, const float* gamma, const float* res, int res_mod,
                     const float* ln_w, const float* ln_b, void* ln_planes, float* ln_f32) {
        GemmParams g = {};
        g.a_pl = a_pl; g.w_pl = w_pl; g.bias = bias; g.gamma = gamma; g.res = res; g.res_mod = res_mod;
        g.C = x; g.M = rows; g.N = dim; g.K = K; g.lda = K; g.ldw = K; g.ldc = dim; g.ldres = dim;
        g.epilogue = EPI_BIAS_LS_RES;
        g.ln_w = ln_w; g.ln_b = ln_b; g.ln_eps = eps; g.ln_planes = ln_planes; g.ln_f32 = ln_f32;
        g.range_flag = range_flag;
        return pope_launch_gemm_rowln(g, stream);
    };
    POPE_MARK(POPE_K_PATCH_EMBED);
    if (fused) {   // `big` is free here: it holds the im2col planes
        const int kp = (3 * w->patch * w->patch + 31) & ~31;
        if (workspace_bytes - size_t(reinterpret_cast<char*>(big) - ws) < size_t(rows) * kp * 4) return POPE_ERR_WORKSPACE;
        POPE_TRY(pope_launch_im2col_planes(img, big, B, H, W, w->patch, kp, range_flag, stream));
        const pope_vit_block_weights& k0 = w->blocks_host[0];
        POPE_TRY(rowln(big, w->patch_wp, kp, nullptr, nullptr, posb, ntok, k0.norm1_w, k0.norm1_b, xn_pl, nullptr));
    } else if (planes && w->patch_wp) {
        POPE_TRY(pope_patch_embed_planes_f32(img, w->patch_wp, posb, x, B, H, W, w->patch, dim, big,
                                             workspace_bytes - size_t(reinterpret_cast<char*>(big) - ws), range_flag, stream));
    } else {
        POPE_TRY(pope_patch_embed_f32(img, w->patch_w, posb, x, B, H, W, w->patch, dim, stream));
    }
    for (int i = 0; i < w->depth; ++i) {
        const pope_vit_block_weights& k = w->blocks_host[i];
        const bool last = i + 1 == w->depth;
        // x = x + ls1(attn(norm1(x)))                                      block.py:105
        if (plain) {   // the same seven launches per block in single-product f16 arithmetic
            POPE_MARK(POPE_K_LAYERNORM);
            POPE_TRY(pope_launch_layernorm_f16(x, k.norm1_w, k.norm1_b, xn_pl, rows, dim, eps, range_flag, stream));
            POPE_MARK(POPE_K_GEMM_QKV);
            // q (pre-scaled by head_dim^-0.5 log2 e), k, v leave the QKV epilogue as f16 rows: the attention kernel stages K / V
            // memory -> LDS directly (attention_f16.hip)
            POPE_TRY(plain_gemm(xn_pl, k.qkv_wp, k.qkv_b, nullptr, qkv, 3 * dim, dim, EPI_QKV_F16, nullptr, nullptr));
            POPE_MARK(POPE_K_ATTENTION);
            POPE_TRY(pope_launch_attention_f16_dma(qkv, att, B, ntok, w->heads, stream));
            POPE_MARK(POPE_K_GEMM_PROJ);
            POPE_TRY(plain_gemm(att, k.proj_wp, k.proj_b, x, nullptr, dim, dim, EPI_BIAS_LS_RES, k.ls1, x));
            POPE_MARK(POPE_K_LAYERNORM);
            POPE_TRY(pope_launch_layernorm_f16(x, k.norm2_w, k.norm2_b, xn_pl, rows, dim, eps, range_flag, stream));
            POPE_MARK(POPE_K_GEMM_FC1);
            POPE_TRY(plain_gemm(xn_pl, k.fc1_wp, k.fc1_b, nullptr, hid_pl, hidden, dim, EPI_BIAS_GELU, nullptr, nullptr));
            POPE_MARK(POPE_K_GEMM_FC2);
            POPE_TRY(plain_gemm(hid_pl, k.fc2_wp, k.fc2_b, x, nullptr, dim, hidden, EPI_BIAS_LS_RES, k.ls2, x));
            for (int t = 0; t < n_taps; ++t)
                if (tap_blocks_host[t] == i && tap_out_host[t]) {
                    POPE_MARK(POPE_K_TAP_COPY);
                    if (hipMemcpyAsync(tap_out_host[t], x, size_t(rows) * dim * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
                        return POPE_ERR_LAUNCH;
                }
            continue;
        }
        if (!fused) {
            POPE_MARK(POPE_K_LAYERNORM);
            if (small) POPE_TRY(pope_launch_layernorm_rowln_order(x, k.norm1_w, k.norm1_b, xn_pl, nullptr, rows, eps, range_flag, stream));
            else if (planes) POPE_TRY(pope_launch_layernorm_planes(x, dim, k.norm1_w, k.norm1_b, xn_pl, rows, dim, eps, range_flag, stream));
            else POPE_TRY(pope_launch_layernorm_f32(x, dim, k.norm1_w, k.norm1_b, xn, dim, rows, dim, eps, stream));
        }
        POPE_MARK(POPE_K_GEMM_QKV);
        if (planes)  // q, k, v stay planes from the QKV epilogue to the attention kernel's LDS
            POPE_TRY(pope_linear_planes_f32(xn_pl, k.qkv_wp, k.qkv_b, nullptr, qkv, rows, 3 * dim, dim, EPI_BIAS, nullptr, nullptr,
                                            range_flag, stream));
        else POPE_TRY(pope_linear_prec_f32(xn, k.qkv_w, k.qkv_b, qkv, rows, 3 * dim, dim, EPI_BIAS, nullptr, nullptr, f32, nullptr, stream));
        POPE_MARK(POPE_K_ATTENTION);
        if (planes) POPE_TRY(pope_launch_attention_f16x3_planes_io(qkv, att, B, ntok, w->heads, stream));
        else POPE_TRY(pope_attention_prec_f32(qkv, att, B, ntok, w->heads, f32, nullptr, stream));
        POPE_MARK(POPE_K_GEMM_PROJ);
        if (fused)
            POPE_TRY(rowln(att, k.proj_wp, dim, k.proj_b, k.ls1, x, 0, k.norm2_w, k.norm2_b, xn_pl, nullptr));
        else if (planes)
            POPE_TRY(pope_linear_planes_f32(att, k.proj_wp, k.proj_b, x, nullptr, rows, dim, dim, EPI_BIAS_LS_RES, k.ls1, x, nullptr, stream));
        else POPE_TRY(pope_linear_prec_f32(att, k.proj_w, k.proj_b, x, rows, dim, dim, EPI_BIAS_LS_RES, k.ls1, x, f32, nullptr, stream));
        // x = x + ls2(mlp(norm2(x)))                                       block.py:106
        if (!fused) {
            POPE_MARK(POPE_K_LAYERNORM);
            if (small) POPE_TRY(pope_launch_layernorm_rowln_order(x, k.norm2_w, k.norm2_b, xn_pl, nullptr, rows, eps, range_flag, stream));
            else if (planes) POPE_TRY(pope_launch_layernorm_planes(x, dim, k.norm2_w, k.norm2_b, xn_pl, rows, dim, eps, range_flag, stream));
            else POPE_TRY(pope_launch_layernorm_f32(x, dim, k.norm2_w, k.norm2_b, xn, dim, rows, dim, eps, stream));
        }
        POPE_MARK(POPE_K_GEMM_FC1);
        if (planes)
            POPE_TRY(pope_linear_planes_f32(xn_pl, k.fc1_wp, k.fc1_b, nullptr, hid_pl, rows, hidden, dim, EPI_BIAS_GELU, nullptr,
                                            nullptr, range_flag, stream));
        else POPE_TRY(pope_linear_prec_f32(xn, k.fc1_w, k.fc1_b, hid, rows, hidden, dim, EPI_BIAS_GELU, nullptr, nullptr, f32, nullptr, stream));
        POPE_MARK(POPE_K_GEMM_FC2);
        if (fused && !last) {
            const pope_vit_block_weights& kn = w->blocks_host[i + 1];
            POPE_TRY(rowln(hid_pl, k.fc2_wp, hidden, k.fc2_b, k.ls2, x, 0, kn.norm1_w, kn.norm1_b, xn_pl, nullptr));
        } else if (fused && x_norm) {   // last block: the final norm (vision_transformer.py:230) as fp32
            POPE_TRY(rowln(hid_pl, k.fc2_wp, hidden, k.fc2_b, k.ls2, x, 0, w->norm_w, w->norm_b, nullptr, x_norm));
        } else if (planes) {
            POPE_TRY(pope_linear_planes_f32(hid_pl, k.fc2_wp, k.fc2_b, x, nullptr, rows, dim, hidden, EPI_BIAS_LS_RES, k.ls2, x,
                                            nullptr, stream));
        } else {
            POPE_TRY(pope_linear_prec_f32(hid, k.fc2_w, k.fc2_b, x, rows, dim, hidden, EPI_BIAS_LS_RES, k.ls2, x, f32, nullptr, stream));
        }
        for (int t = 0; t < n_taps; ++t)
            if (tap_blocks_host[t] == i && tap_out_host[t]) {
                POPE_MARK(POPE_K_TAP_COPY);
                if (hipMemcpyAsync(tap_out_host[t], x, size_t(rows) * dim * sizeof(float), hipMemcpyDeviceToDevice,
                                   stream) != hipSuccess)
                    return POPE_ERR_LAUNCH;
            }
    }
    if (x_norm && !fused) {
        POPE_MARK(POPE_K_LAYERNORM);
        if (small) POPE_TRY(pope_launch_layernorm_rowln_order(x, w->norm_w, w->norm_b, nullptr, x_norm, rows, eps, nullptr, stream));
        else POPE_TRY(pope_launch_layernorm_f32(x, dim, w->norm_w, w->norm_b, x_norm, dim, rows, dim, eps, stream));
    }
    POPE_MARK(-1);  // closing event
#undef POPE_TRY
#undef POPE_MARK
    return POPE_OK;
}


int pope_vit_forward_f32(const pope_vit_weights* w, const float* img, int B, int H, int W, const float* posb,
                         float* x_prenorm, float* x_norm, int n_taps, const int* tap_blocks_host,
                         float* const* tap_out_host, void* workspace, size_t workspace_bytes, unsigned* range_flag,
                         void* stream) {
    Recorder rec{nullptr, 0, nullptr, 0};
    return vit_forward_impl(w, img, B, H, W, posb, x_prenorm, x_norm, n_taps, tap_blocks_host, tap_out_host, workspace,
                            workspace_bytes, range_flag, stream, rec);
}

int pope_vit_forward_profiled_mask_f32(const pope_vit_weights* w, const float* img, int B, int H, int W,
                                       const float* posb, float* x_prenorm, float* x_norm, void* workspace,
                                       size_t workspace_bytes, unsigned* range_flag, void* stream,
                                       void* const* events_host, int n_events, int* kinds_host, int* n_launches_host,
                                       unsigned kind_mask) {
    if (!events_host || n_events < 2 || !kinds_host || !n_launches_host) return POPE_ERR_ARG;
    Recorder rec{events_host, n_events, kinds_host, 0};
    rec.mask = kind_mask;
    const int rc = vit_forward_impl(w, img, B, H, W, posb, x_prenorm, x_norm, 0, nullptr, nullptr, workspace,
                                    workspace_bytes, range_flag, stream, rec);
    *n_launches_host = rec.n > 0 ? rec.n - 1 : 0;
    return rc;
}

int pope_vit_launch_count(int depth) { return depth > 0 ? 7 * depth + 2 : 0; }

int pope_event_create(void** event_host) {
    if (!event_host) return POPE_ERR_ARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return POPE_ERR_LAUNCH;
    *event_host = e;
    return POPE_OK;
}
int pope_event_destroy(void* event) {
    return hipEventDestroy(static_cast<hipEvent_t>(event)) == hipSuccess ? POPE_OK : POPE_ERR_LAUNCH;
}
int pope_event_elapsed_ms(void* start, void* stop, float* ms_host) {
    if (!ms_host) return POPE_ERR_ARG;
    return hipEventElapsedTime(ms_host, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) == hipSuccess
               ? POPE_OK : POPE_ERR_LAUNCH;
}

namespace {
// workspace carving of the dense matcher, shared by the size query and the launcher
struct MatchLayout {
    size_t nl, ns, part, rowp, colp, pl0, pl1, simb, total;
    int ncb, nrb, nrb2, ldp;
    MatchLayout(int n, int L, int S, int C, int precision, bool publish_conf) {
        nl = align_up(size_t(n) * L * 4, 256);
        ns = align_up(size_t(n) * S * 4, 256);
        ncb = 2 * ((S + 127) / 128);
        nrb = 4 * ((L + 127) / 128);
        nrb2 = pope_match_nrb2(L);
        ldp = (S + 3) & ~3;
        part = align_up(size_t(n) * nrb2 * ldp * 4, 256);
        const bool x3 = precision == POPE_PREC_F16X3;
        rowp = x3 ? align_up(size_t(n) * L * ncb * 8, 256) : 0;
        colp = x3 ? align_up(size_t(n) * nrb * ldp * 4, 256) : 0;
        pl0 = x3 ? align_up(size_t(n) * L * C * 4, 256) : 0;
        pl1 = x3 ? align_up(size_t(n) * S * C * 4, 256) : 0;
        simb = publish_conf ? 0 : align_up(size_t(n) * L * S * 4, 256);
        total = 7 * nl + 3 * ns + part + rowp + 2 * colp + pl0 + pl1 + simb;
    }
};
}  // namespace

size_t pope_dense_match_workspace_bytes(int n, int L, int S) {
    if (n <= 0 || L <= 0 || S <= 0) return 0;
    return MatchLayout(n, L, S, 4, POPE_PREC_F32_MFMA, true).total;
}

size_t pope_dense_match_workspace_bytes_prec(int n, int L, int S, int C, int precision, int publish_conf) {
    if (n <= 0 || L <= 0 || S <= 0 || C <= 0) return 0;
    return MatchLayout(n, L, S, C, precision, publish_conf != 0).total;
}

int pope_dense_match_f32(const float* feat0, long long stride0, const float* feat1, long long stride1, int n, int L,
                         int S, int C, int h0, int w0, int h1, int w1, float thr, int border_rm, float temperature, float scale, float* conf_matrix,
                         long long* b_ids, long long* i_ids, long long* j_ids, float* mconf, float* mkpts0_c,
                         float* mkpts1_c, int* counts, void* workspace, size_t workspace_bytes, void* stream) {
    return pope_dense_match_prec_f32(feat0, stride0, feat1, stride1, n, L, S, C, h0, w0, h1, w1, thr, border_rm, temperature, scale,
                                     conf_matrix, b_ids, i_ids, j_ids, mconf, mkpts0_c, mkpts1_c, counts, workspace,
                                     workspace_bytes, POPE_PREC_F32_MFMA, nullptr, stream);
}

int pope_dense_match_prec_f32(const float* feat0, long long stride0, const float* feat1, long long stride1, int n, int L,
                              int S, int C, int h0, int w0, int h1, int w1, float thr, int border_rm, float temperature, float scale, float* conf_matrix,
                              long long* b_ids, long long* i_ids, long long* j_ids, float* mconf, float* mkpts0_c,
                              float* mkpts1_c, int* counts, void* workspace, size_t workspace_bytes, int precision,
                              unsigned* range_flag, void* stream) {
    StreamDevice on_device(stream);
    if (!feat0 || !feat1 || !b_ids || !i_ids || !j_ids || !mconf || !mkpts0_c || !mkpts1_c || !counts || !workspace)
        return POPE_ERR_ARG;
    if (n <= 0 || L <= 0 || S <= 0 || C <= 0) return POPE_ERR_ARG;
    if (precision != POPE_PREC_F32_MFMA && precision != POPE_PREC_F16X3) return POPE_ERR_ARG;
    const bool publish = conf_matrix != nullptr;
    const MatchLayout lay(n, L, S, C, precision, publish);
    if (workspace_bytes < lay.total) return POPE_ERR_WORKSPACE;
    char* ws = static_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* q = ws; ws += bytes; return q; };
    MatchParams p = {};
    p.feat0 = feat0; p.feat1 = feat1;
    p.n = n; p.L = L; p.S = S; p.C = C;
    p.bs0 = stride0; p.bs1 = stride1;
    p.h0 = h0; p.w0 = w0; p.h1 = h1; p.w1 = w1;
    p.thr = thr; p.temperature = temperature; p.border = border_rm; p.scale = scale;
    p.publish_conf = publish;
    p.ncb = lay.ncb; p.nrb = lay.nrb; p.nrb2 = lay.nrb2; p.ldp = lay.ldp;
    p.row_max = reinterpret_cast<float*>(take(lay.nl));
    p.row_sum = reinterpret_cast<float*>(take(lay.nl));
    p.conf_rowmax = reinterpret_cast<float*>(take(lay.nl));
    p.row_j = reinterpret_cast<int*>(take(lay.nl));
    p.row_conf = reinterpret_cast<float*>(take(lay.nl));
    p.row_arg = reinterpret_cast<int*>(take(lay.nl));
    p.row_cnt = reinterpret_cast<int*>(take(lay.nl));
    p.col_max = reinterpret_cast<float*>(take(lay.ns));
    p.col_sum = reinterpret_cast<float*>(take(lay.ns));
    p.conf_colmax = reinterpret_cast<float*>(take(lay.ns));
    p.colmax_part = reinterpret_cast<float*>(take(lay.part));
    if (precision == POPE_PREC_F16X3) {
        p.row_part = reinterpret_cast<float*>(take(lay.rowp));
        p.col_pmax = reinterpret_cast<float*>(take(lay.colp));
        p.col_psum = reinterpret_cast<float*>(take(lay.colp));
        p.planes0 = take(lay.pl0);
        p.planes1 = take(lay.pl1);
    }
    p.sim = publish ? conf_matrix : reinterpret_cast<float*>(take(lay.simb));
    p.counts = counts;
    p.range_flag = range_flag;
    p.b_ids = b_ids; p.i_ids = i_ids; p.j_ids = j_ids;
    p.mconf = mconf; p.mkpts0 = mkpts0_c; p.mkpts1 = mkpts1_c;
    return pope_launch_dense_match_f32(p, static_cast<hipStream_t>(stream));
}

size_t pope_loftr_layer_workspace_bytes(int n, int L, int S, int C, int nhead) {
    if (n <= 0 || L <= 0 || S <= 0 || C <= 0 || nhead <= 0) return 0;
    return pope_loftr_layer_workspace(n, L, S, C, nhead);
}

int pope_loftr_encoder_layer_f32(const pope_loftr_layer_weights* w, float* x, const float* source, int n, int L, int S, int C,
                                 int nhead, float ln_eps, int precision, void* workspace, size_t workspace_bytes,
                                 unsigned* range_flag, void* stream) {
    StreamDevice on_device(stream);
    if (!w) return POPE_ERR_ARG;
    LoftrLayerParams p = {};
    p.x = x; p.source = source; p.n = n; p.L = L; p.S = S; p.C = C; p.H = nhead;
    p.precision = precision;
    if (precision == POPE_PREC_F32_MFMA) {   // the five `*_wp` are then plain fp32 [out, in] matrices
        p.q_w = static_cast<const float*>(w->q_wp); p.kv_w = static_cast<const float*>(w->kv_wp);
        p.merge_w = static_cast<const float*>(w->merge_wp); p.mlp0_w = static_cast<const float*>(w->mlp0_wp);
        p.mlp1_w = static_cast<const float*>(w->mlp1_wp);
    }
    p.q_wp = w->q_wp; p.kv_wp = w->kv_wp; p.merge_wp = w->merge_wp; p.mlp0_wp = w->mlp0_wp; p.mlp1_wp = w->mlp1_wp;
    p.norm1_w = w->norm1_w; p.norm1_b = w->norm1_b; p.norm2_w = w->norm2_w; p.norm2_b = w->norm2_b;
    p.ln_eps = ln_eps; p.ws = workspace; p.ws_bytes = workspace_bytes; p.range_flag = range_flag;
    return pope_launch_loftr_layer(p, static_cast<hipStream_t>(stream));
}

size_t pope_resnetfpn_workspace_bytes(int n, int H, int W) {
    if (n <= 0 || H < 16 || W < 16 || (H & 7) || (W & 7)) return 0;
    return pope_resnetfpn_workspace(n, H, W);
}

int pope_resnetfpn_forward_f32(const pope_resnetfpn_weights* w, const float* gray, int n, int H, int W, int precision, float* out_c,
                               float* out_f, void* workspace, size_t workspace_bytes, unsigned* range_flag, void* stream) {
    StreamDevice on_device(stream);
    if (!w) return POPE_ERR_ARG;
    ResnetFpnParams q = {};
    q.img = gray; q.n = n; q.H = H; q.W = W;
    q.precision = precision;
    for (int i = 0; i < 22; ++i) { q.w[i] = w->w[i]; q.b[i] = w->b[i]; q.wf[i] = static_cast<const float*>(w->w[i]); }
    q.out_c = out_c; q.out_f = out_f; q.ws = workspace; q.ws_bytes = workspace_bytes; q.range_flag = range_flag;
    return pope_launch_resnetfpn(q, static_cast<hipStream_t>(stream));
}

size_t pope_fine_preprocess_workspace_bytes(int M, int Wn, int Cc, int Cf) {
    if (M <= 0 || Wn <= 0 || Cc <= 0 || Cf <= 0) return 0;
    return pope_fine_preprocess_workspace(M, Wn * Wn, Cc, Cf);
}

int pope_fine_preprocess_f32(const float* feat_f0, const long long* strides0, int H0, int W0, int wc0, const float* feat_f1,
                             const long long* strides1, int H1, int W1, int wc1, const float* feat_c0, const float* feat_c1, int L,
                             int S, int Cc, int Cf, const long long* b_ids, const long long* i_ids, const long long* j_ids, int M, int Wn,
                             int stride, const void* down_wp, const float* down_b, const void* merge_wp, const float* merge_b,
                             int precision, float* out, void* workspace, size_t workspace_bytes, unsigned* range_flag, void* stream) {
    StreamDevice on_device(stream);
    if (!strides0 || !strides1 || H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0 || wc0 <= 0 || wc1 <= 0 || L <= 0 || S <= 0) return POPE_ERR_ARG;
    FinePreParams q = {};
    q.f0 = feat_f0; q.f1 = feat_f1;
    for (int i = 0; i < 4; ++i) { q.s0[i] = strides0[i]; q.s1[i] = strides1[i]; }
    q.H0 = H0; q.W0 = W0; q.H1 = H1; q.W1 = W1; q.wc0 = wc0; q.wc1 = wc1;
    q.fc0 = feat_c0; q.fc1 = feat_c1; q.L = L; q.S = S; q.Cc = Cc; q.Cf = Cf;
    q.b_ids = b_ids; q.i_ids = i_ids; q.j_ids = j_ids; q.M = M; q.Wn = Wn; q.stride = stride;
    q.down_wp = down_wp; q.down_b = down_b; q.merge_wp = merge_wp; q.merge_b = merge_b;
    q.precision = precision;
    q.down_w = static_cast<const float*>(down_wp); q.merge_w = static_cast<const float*>(merge_wp);
    q.out = out; q.ws = workspace; q.ws_bytes = workspace_bytes; q.range_flag = range_flag;
    return pope_launch_fine_preprocess(q, static_cast<hipStream_t>(stream));
}

int pope_fine_match_f32(const float* win0, const float* win1, int M, int Wn, int C, const float* mkpts1_c, float scale_px,
                        float* expec_f, float* mkpts1_f, void* stream) {
    StreamDevice on_device(stream);
    return pope_launch_fine_match(win0, win1, M, Wn, C, mkpts1_c, scale_px, expec_f, mkpts1_f, static_cast<hipStream_t>(stream));
}

static bool sam_params(const pope_sam_encoder_weights* w, int B, SamEncParams& q, SamBlockParams* blocks) {
    if (!w || !w->blocks_host || w->depth <= 0 || w->depth > 64) return false;
    q = SamEncParams{};
    q.B = B; q.img = w->img; q.patch = w->patch; q.dim = w->dim; q.depth = w->depth; q.heads = w->heads; q.hidden = w->hidden;
    q.out_chans = w->out_chans; q.window = w->window; q.precision = w->precision;
    q.block_eps = w->block_eps; q.neck_eps = w->neck_eps;
    q.patch_wp = w->patch_wp; q.patch_b = w->patch_b; q.pos = w->pos; q.ones = w->ones;
    q.neck0_wp = w->neck0_wp; q.neck1_w = w->neck1_w; q.neck1_b = w->neck1_b; q.neck2_wp = w->neck2_wp;
    q.neck3_w = w->neck3_w; q.neck3_b = w->neck3_b;
    for (int i = 0; blocks && i < w->depth; ++i) {
        const pope_sam_block_weights& s = w->blocks_host[i];
        SamBlockParams& d = blocks[i];
        d.norm1_w = s.norm1_w; d.norm1_b = s.norm1_b; d.qkv_wp = s.qkv_wp; d.qkv_b = s.qkv_b; d.proj_wp = s.proj_wp;
        d.proj_b = s.proj_b; d.rel_h = s.rel_h; d.rel_w = s.rel_w; d.norm2_w = s.norm2_w; d.norm2_b = s.norm2_b;
        d.fc1_wp = s.fc1_wp; d.fc1_b = s.fc1_b; d.fc2_wp = s.fc2_wp; d.fc2_b = s.fc2_b; d.global = s.global_attn;
    }
    q.blocks = blocks;
    return true;
}

size_t pope_sam_encoder_workspace_bytes(const pope_sam_encoder_weights* w, int B) {
    SamEncParams q;
    if (!sam_params(w, B, q, nullptr)) return 0;
    return pope_sam_encoder_workspace(q);
}

int pope_sam_encoder_forward_f32(const pope_sam_encoder_weights* w, const float* image, int B, float* out, int n_taps,
                                 const int* tap_blocks_host, float* const* tap_out_host, void* workspace, size_t workspace_bytes,
                                 unsigned* range_flag, void* stream) {
    StreamDevice on_device(stream);
    SamEncParams q;
    SamBlockParams blocks[64];
    if (!sam_params(w, B, q, blocks)) return POPE_ERR_ARG;
    if (n_taps < 0 || (n_taps > 0 && (!tap_blocks_host || !tap_out_host))) return POPE_ERR_ARG;
    q.image = image; q.out = out;
    q.n_taps = n_taps; q.tap_blocks = tap_blocks_host; q.tap_out = tap_out_host;
    q.ws = workspace; q.ws_bytes = workspace_bytes; q.range_flag = range_flag;
    return pope_launch_sam_encoder(q, static_cast<hipStream_t>(stream));
}

int pope_preprocess_u8_f32(const unsigned char* img_hwc, int P, int Hin, int Win, const int* hstart, const int* hcount,
                           const int* hk, int kh, const int* vstart, const int* vcount, const int* vk, int kv, int top, int left,
                           int ch, int cw, int row0, int nrows, const float* mean_host, const float* std_host, float* out,
                           unsigned char* scratch, size_t scratch_bytes, void* stream) {
    StreamDevice on_device(stream);
    if (!mean_host || !std_host || P <= 0 || nrows <= 0 || cw <= 0) return POPE_ERR_ARG;
    if (scratch_bytes < size_t(P) * nrows * cw * 3) return POPE_ERR_WORKSPACE;
    PreprocParams p = {};
    p.img = img_hwc; p.P = P; p.Hin = Hin; p.Win = Win;
    p.hstart = hstart; p.hcount = hcount; p.hk = hk; p.kh = kh;
    p.vstart = vstart; p.vcount = vcount; p.vk = vk; p.kv = kv;
    p.top = top; p.left = left; p.ch = ch; p.cw = cw; p.row0 = row0; p.nrows = nrows;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean_host[c]; p.std[c] = std_host[c]; }
    p.tmp = scratch; p.out = out;
    return pope_launch_preprocess(p, static_cast<hipStream_t>(stream));
}

int pope_crop_normalize_u8_f32(const unsigned char* img_hwc, int P, int Hin, int Win, int top, int left, int ch, int cw,
                               const float* mean_host, const float* std_host, float* out, void* stream) {
    StreamDevice on_device(stream);
    return pope_launch_crop_norm(img_hwc, P, Hin, Win, top, left, ch, cw, mean_host, std_host, out, static_cast<hipStream_t>(stream));
}

int pope_gray_u8_f32(const unsigned char* bgr_hwc, int P, int H, int W, float* out, void* stream) {
    StreamDevice on_device(stream);
    if (P <= 0 || H <= 0 || W <= 0) return POPE_ERR_ARG;
    return pope_launch_gray(bgr_hwc, size_t(P) * H * W, out, static_cast<hipStream_t>(stream));
}

int pope_crop_warp_u8(const unsigned char* img_hwc, int H, int W, int C, const double* minv, const int* win, int P, int oh, int ow,
                      unsigned char* out, void* stream) {
    StreamDevice on_device(stream);
    return pope_launch_crop_warp(img_hwc, H, W, C, minv, win, P, oh, ow, out, static_cast<hipStream_t>(stream));
}

int pope_streaming_top3_host(const float* scores, int P, float* slot_scores, long long* slot_index) {
    if (!scores || !slot_scores || !slot_index || P < 0) return POPE_ERR_ARG;
    for (int k = 0; k < 3; ++k) { slot_scores[k] = 0.f; slot_index[k] = -1; }
    for (int p = 0; p < P; ++p) {
        const float s = scores[p];
        if (s > slot_scores[0] || s > slot_scores[1] || s > slot_scores[2]) {
            int k = 0;  // np.argmin: first minimum
            if (slot_scores[1] < slot_scores[k]) k = 1;
            if (slot_scores[2] < slot_scores[k]) k = 2;
            slot_scores[k] = s;
            slot_index[k] = p;
        }
    }
    return POPE_OK;
}

}  // extern "C"

size_t pope_estimate_pose_workspace_bytes(int B, long long M) { return pope_pose_workspace(B, M); }

int pope_estimate_pose_f64(const float* kpts0, const float* kpts1, const int* counts, const double* K0, const double* K1, int B,
                           long long M, double thresh, double conf, int max_iters, unsigned long long seed, double* R, double* t,
                           double* E, unsigned char* inliers, int* info, void* workspace, size_t workspace_bytes, void* stream) {
    StreamDevice on_device(stream);
    PoseParams q = {};
    q.kpts0 = kpts0; q.kpts1 = kpts1; q.counts = counts; q.K0 = K0; q.K1 = K1; q.B = B; q.M = M;
    q.thresh = thresh; q.conf = conf; q.max_iters = max_iters; q.seed = seed;
    q.R = R; q.t = t; q.E = E; q.inliers = inliers; q.info = info;
    return pope_launch_estimate_pose(q, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int pope_five_point_f64(const double* x0, const double* x1, int S, double* E_out, int* n_out, void* stream) {
    StreamDevice on_device(stream);
    return pope_launch_five_point(x0, x1, S, E_out, n_out, static_cast<hipStream_t>(stream));
}
