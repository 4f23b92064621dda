// SAM image encoder (segment_anything/segment_anything/modeling/image_encoder.py:17-118: ViT-B/L/H with 14x14 window
// attention, four global blocks, decomposed relative position terms, and the 1x1 / 3x3 convolution neck) on the f16x3
// planes GEMMs of the DINOv2 path.  BASELINE config 5 / SURVEY.md §8 f-3.
//
// What is new here is the attention.  softmax(scale q.k^T + q.Rh[qh,kh] + q.Rw[qw,kw]) (image_encoder.py:225-231,
// 325-358) is evaluated as ONE matrix product per (window, head) by widening the operands:
//     Q'[n] = [ scale q[n] | q[n].Rh[qh(n), 0..KH) | q[n].Rw[qw(n), 0..KW) ]      (all times log2 e)
//     K'[m] = [ k[m]       | onehot(kh(m))          | onehot(kw(m))          ]
// so Q'.K'^T is the biased score and the flash kernel needs no bias path at all: the relative-position terms ride on
// the matrix cores (K' one-hot columns are exact in f16 and have no lo plane: 2 MFMAs per step there instead of 3).
// The 64 x 64 global blocks skip the widening (round 4, template flag BIAS): their 32-key tiles lie inside one key row, so the
// two terms are a per-lane register table plus one LDS broadcast that the score accumulators start from.
// The QKV GEMM's epilogue (gemm_planes.hip, EPI_SAM_QKV), `sam_pad_tokens_kernel` and `sam_attn_relpos_kernel` build Q', K', V
// as f16 hi/lo planes per (window, head) — the window partition is a row map of the epilogue, and the zero-padded tokens of the bottom / right windows
// (image_encoder.py:251-254, padded AFTER norm1) get k = v = the qkv bias, exactly what Linear(0) gives the reference.
// `sam_attn_kernel` is the single-stage f16x3 flash kernel of attention_f16x3.hip re-cut for 32-key tiles, a
// K depth of 16 * NSTEP and 32 * DVT value columns; its epilogue un-partitions (drops the pad queries) and writes the
// activation planes of the proj GEMM.  (Two LDS stages with one barrier per tile were measured and dropped: global
// blocks 0.83 -> 0.87 ms, window blocks 0.082 -> 0.080 ms in the f16 mode; the tile is bound by its own MFMA + softmax
// chain at two waves per SIMD, not by the staging.)
// Two precisions (pope_hip.h): POPE_PREC_F16X3 as above; POPE_PREC_F16 = plain f16 operands, one MFMA per product
// (template flag PLAIN here and in gemm_planes.hip), fp32 accumulators / softmax / LayerNorm / residual stream in both.
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr float L2E = 1.44269504088896340736f;
constexpr float A_SCALE = K_PLANES_ACT_SCALE;

inline int grid_for(long long total, int per_block = 256) {
    long long b = (total + per_block - 1) / per_block;
    const long long cap = 64ll * pope_cu_count();
    return int(b < 1 ? 1 : (b > cap ? cap : b));
}
inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

__device__ __forceinline__ f16x8 cat(f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---- patch embed operand: image [B, 3, S, S] -> activation planes [B * g * g, 3 * P * P], k = (c, ky, kx) as
// Conv2d's weight.reshape(dim, -1) (image_encoder.py:385-393); P % 8 == 0
template <bool PLAIN>
__global__ __launch_bounds__(256) void sam_im2col_kernel(const float* __restrict__ img, _Float16* __restrict__ out, int B, int S,
                                                         int P, unsigned* range_flag) {
    const int g = S / P, K = 3 * P * P, pieces = K / 8;
    const long long total = (long long)B * g * g * pieces;
    float amax = 0.f;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int pc = int(id % pieces);
        const long long row = id / pieces;
        const int px = int(row % g), py = int((row / g) % g), b = int(row / ((long long)g * g));
        const int k = pc * 8, c = k / (P * P), ky = (k - c * P * P) / P, kx = k - c * P * P - ky * P;
        const float* src = img + (((size_t)b * 3 + c) * S + (py * P + ky)) * S + px * P + kx;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
        amax = pope_amax4(pope_amax4(amax, v0), v1);
        const float s8 = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
        if (!(s8 == s8)) amax = INFINITY;   // NaN (fmax drops it)
        if constexpr (PLAIN) {   // f16 row-major
            *reinterpret_cast<f16x8*>(out + (size_t)row * K + k) =
                cat(__builtin_convertvector(v0 * A_SCALE, f16x4), __builtin_convertvector(v1 * A_SCALE, f16x4));
        } else {
            f16x4 h0, l0, h1, l1;
            pope_split4(v0 * A_SCALE, h0, l0);
            pope_split4(v1 * A_SCALE, h1, l1);
            _Float16* o = out + (size_t)row * 2 * K + (k >> 5) * 64 + (k & 31);
            *reinterpret_cast<f16x8*>(o) = cat(h0, h1);
            *reinterpret_cast<f16x8*>(o + 32) = cat(l0, l1);
        }
    }
    pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax * A_SCALE < POPE_F16_OVERFLOW));
}

// ---- attention geometry (host and device) -----------------------------------------------------------------------
struct AttnGeom {
    int B, g;             // images, token grid (g x g)
    int ws, nw;           // window side and windows per axis (global block: ws = g, nw = 1)
    int heads, hd, dim;   // hd = 64 or 80
    int Nq, Npad;         // tokens per window (ws * ws) and rounded up to the 32-key tile
    int DQ, HDP, DV;      // score depth (16 * NSTEP >= hd + 2 ws), lo-plane depth (= hd), value columns (32 * DVT)
};

// Operand planes of one block's attention (halves; G = B nw^2 heads groups, n = token inside its window):
//   Qp [G][Npad][DQ hi | DQ lo], Kp [G][Npad][DQ hi | hd lo], Vp [G][Npad][DV hi | DV lo]     (PLAIN: the hi parts only)
// Who writes what: the QKV GEMM's epilogue (gemm_planes.hip, EPI_SAM_QKV) writes q * scale * log2 e, k and v of every real
// token into its rows — the window partition is a row map; `sam_pad_tokens_kernel` writes the rows of the zero-padded
// tokens of the edge windows (image_encoder.py:251-254: padded AFTER norm1, so their q, k, v are the qkv bias);
// `sam_attn_relpos_kernel` adds the relative-position columns of Q'; everything that does not depend on the block
// (zero rows n >= Nq, zero value columns hd..DV, K's one-hot columns, the row map) is written once per forward pass.
template <bool PLAIN>
__global__ __launch_bounds__(256) void sam_pad_tokens_kernel(const float* __restrict__ qkv_bias, _Float16* __restrict__ Qp,
                                                             _Float16* __restrict__ Kp, _Float16* __restrict__ Vp, AttnGeom a,
                                                             unsigned* range_flag) {
    // row pitches (halves): PLAIN rows carry no lo halves
    const int q_row = PLAIN ? a.DQ : 2 * a.DQ, k_row = PLAIN ? a.DQ : a.DQ + a.HDP, v_row = PLAIN ? a.DV : 2 * a.DV;
    const int hp = a.hd / 8;   // 8-column pieces of q | k | v per (group, token)
    // the pad tokens of one image, enumerated: the bottom strip (rows g .. gp of the padded gp x gp grid), then the right
    // strip of the rows above it
    const int gp = a.nw * a.ws, pr = gp - a.g, n_pad = gp * gp - a.g * a.g;
    const long long total = (long long)a.B * n_pad * a.heads * 3 * hp;
    const float scale = 1.0f / sqrtf(float(a.hd)) * L2E;   // head_dim ** -0.5 (image_encoder.py:206), log2 domain
    float amax = 0.f;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int piece = int(id % (3 * hp));
        long long rest = id / (3 * hp);
        const int head = int(rest % a.heads);
        rest /= a.heads;
        const int pt = int(rest % n_pad), b = int(rest / n_pad);
        int y, x;
        if (pt < pr * gp) { y = a.g + pt / gp; x = pt - (pt / gp) * gp; }
        else { const int r2 = pt - pr * gp; y = r2 / pr; x = a.g + r2 - (r2 / pr) * pr; }
        const int wy = y / a.ws, wx = x / a.ws;
        const int n = (y - wy * a.ws) * a.ws + (x - wx * a.ws);
        const int grp = ((b * a.nw + wy) * a.nw + wx) * a.heads + head;
        const int which = piece / hp, c0 = 8 * (piece - which * hp);
        const float* src = qkv_bias + which * a.dim + head * a.hd + c0;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
        if (which == 0) { v0 = v0 * scale; v1 = v1 * scale; }
        amax = pope_amax4(pope_amax4(amax, v0), v1);
        const float s8 = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
        if (!(s8 == s8)) amax = INFINITY;
        const size_t row = (size_t)grp * a.Npad + n;
        _Float16* hi_dst;
        _Float16* lo_dst;
        if (which == 0) { hi_dst = Qp + row * q_row + c0; lo_dst = hi_dst + a.DQ; }
        else if (which == 1) { hi_dst = Kp + row * k_row + c0; lo_dst = hi_dst + a.DQ; }
        else { hi_dst = Vp + row * v_row + c0; lo_dst = hi_dst + a.DV; }
        if constexpr (PLAIN) {
            *reinterpret_cast<f16x8*>(hi_dst) = cat(__builtin_convertvector(v0, f16x4), __builtin_convertvector(v1, f16x4));
        } else {
            f16x4 h0, l0, h1, l1;
            pope_split4(v0, h0, l0);
            pope_split4(v1, h1, l1);
            *reinterpret_cast<f16x8*>(hi_dst) = cat(h0, h1);
            *reinterpret_cast<f16x8*>(lo_dst) = cat(l0, l1);
        }
    }
    pope_range_flag(range_flag, POPE_RANGE_QKV, !(amax < POPE_F16_OVERFLOW));
}

// Once per forward pass and geometry (the operand buffers are zero-filled first): the one-hot columns of K'
// (they depend on the token's position in its window only) ...
__global__ __launch_bounds__(256) void sam_onehot_kernel(_Float16* __restrict__ Kp, AttnGeom a, int k_row) {
    const int G = a.B * a.nw * a.nw * a.heads;
    const long long total = (long long)G * a.Nq;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < total; id += 256ll * gridDim.x) {
        const int n = int(id % a.Nq), grp = int(id / a.Nq);
        _Float16* row = Kp + ((size_t)grp * a.Npad + n) * k_row + a.hd;
        row[n / a.ws] = _Float16(1.0f);
        row[a.ws + n % a.ws] = _Float16(1.0f);
    }
}
// ... and the window partition (image_encoder.py:238-259) as a row map for the QKV GEMM's epilogue:
// map[token row t] = (window batch * heads) * Npad + position of the token in its window
__global__ __launch_bounds__(256) void sam_rowmap_kernel(int* __restrict__ map, AttnGeom a) {
    const int rows = a.B * a.g * a.g;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < rows; t += 256 * gridDim.x) {
        const int x = t % a.g, y = (t / a.g) % a.g, b = t / (a.g * a.g);
        const int wy = y / a.ws, wx = x / a.ws;
        map[t] = ((b * a.nw + wy) * a.nw + wx) * a.heads * a.Npad + (y - wy * a.ws) * a.ws + (x - wx * a.ws);
    }
}

// Rh / Rw: [ws][ws][hd] fp32, the gathered tables get_rel_pos returns (image_encoder.py:288-316; host, once per model).
// q.Rh[qh, 0..ws) is the same small matrix product for every token of window row qh (and q.Rw[qw, 0..ws) for every token of
// column qw), so the columns are made on the matrix cores, one wave per (window batch, axis, line r, head group):
//     D[j][rho] = sum_k R[r][j][k] * q[rho][k],    rho = (head, position on the line), 32 of them per MFMA column block
// A = the line's table slice, split into f16 hi / lo (x 256) in registers once per wave; B = q from the Q' rows the QKV
// epilogue (and the pad-token kernel) wrote (q * scale * log2 e as hi [+ lo]).  f16x3: R_lo.q_hi + R_hi.q_lo + R_hi.q_hi;
// f16 mode: q has no lo half.  What the kernel costs is its memory pattern, not the arithmetic (scripts/sam_relpos_lab.hip,
// profiles/r04/sam_relpos_lab.txt: 18 us of 110 without loads and stores), so
//  * a block's 32 q rows are fetched as whole rows by neighbouring lanes (10 lanes x 16 bytes per row) one block ahead
//    and handed to the MFMA layout through a wave-private LDS tile (no barrier: a wave's LDS operations stay in order);
//  * lanes c and c + 32 swap half of their results so that each holds 8 consecutive j of its q row: one 16-byte store;
//  * window blocks walk the task list XCD by XCD (`xcd_remap`): the two axes of a window read the same 0.7 MB of Q'
//    and write the two halves of the same 56-byte segments, which then meet in one L2.
// Rows n >= Nq and columns j >= 2 ws of Q' stay the zeros of the once-per-pass memset.
// (Rounds 2-3 ran this on the vector ALU, one (token, j) pair per thread with the table row in registers.)
typedef unsigned u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte store at dword alignment
// TAB (global blocks in bias mode, `sam_attn_kernel<.., BIAS>`): the columns go to the bias table [G Npad][relh 0..ws | relw 0..ws]
// instead of Q' — fp32 in the f16x3 mode (the accumulators as they are), f16 in the plain-f16 mode.
template <int HD, int MB, bool PLAIN, bool TAB>
__global__ __launch_bounds__(256) void sam_attn_relpos_kernel(const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                              _Float16* __restrict__ Qp, void* __restrict__ bias_tab, AttnGeom a,
                                                              int hpg, int n_tasks, unsigned ws_magic, int by_xcd,
                                                              unsigned* range_flag) {
    constexpr int KS = HD / 16, PR = HD / 8, ST = PR + 1, NT = PR / 2;   // 16-byte pieces per q row, LDS row stride, fetches per lane
    constexpr int NP = PLAIN ? 1 : 2;
    constexpr float W_SCALE = 256.0f;
    __shared__ u32x4 stage_all[4][NP][32 * ST];
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    u32x4(*stage)[32 * ST] = stage_all[threadIdx.x >> 6];
    int task = (by_xcd ? xcd_remap(blockIdx.x, gridDim.x) : int(blockIdx.x)) * 4 + (threadIdx.x >> 6);
    if (task >= n_tasks) return;   // no barrier below
    const int n_hg = a.heads / hpg;
    const int hg = task % n_hg;
    task /= n_hg;
    const int r = task % a.ws;
    task /= a.ws;
    const int axis = task & 1, wb = task >> 1;
    const int q_row = PLAIN ? a.DQ : 2 * a.DQ;

    // A fragments: lane (c, h) holds R[r][j = 32 mb + c][16 s + 8 h + 0..7] * 256 as hi and lo
    f16x8 rh[MB][KS], rl[MB][KS];
    const float* tab = (axis ? Rw : Rh) + (size_t)r * a.ws * HD + 8 * h;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int j = 32 * mb + c;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (j < a.ws) {
                v0 = *reinterpret_cast<const f32x4*>(tab + (size_t)j * HD + 16 * s);
                v1 = *reinterpret_cast<const f32x4*>(tab + (size_t)j * HD + 16 * s + 4);
            }
            f16x4 h0, l0, h1, l1;
            pope_split4(v0 * W_SCALE, h0, l0);
            pope_split4(v1 * W_SCALE, h1, l1);
            rh[mb][s] = cat(h0, h1);
            rl[mb][s] = cat(l0, l1);
        }
    }

    const int n_rho = a.ws * hpg;
    const size_t grp0 = (size_t)(wb * a.heads + hg * hpg) * a.Npad;
    auto row_of = [&](int rho) -> size_t {   // Q' row of the rho-th (head, position) of this line; rho / ws by the host's reciprocal
        const int hl = int(__umulhi(unsigned(rho), ws_magic)), i = rho - hl * a.ws;
        return grp0 + (size_t)hl * a.Npad + (axis ? i * a.ws + r : r * a.ws + i);
    };
    u32x4 pf[NP][NT];
    auto fetch = [&](int rho0) {   // piece p = lane + 64 t of the block: row p / PR, 16-byte piece p % PR
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = lane + 64 * t, cr = p / PR, pc = p - cr * PR;
            const bool in = rho0 + cr < n_rho;
            const _Float16* src = Qp + row_of(rho0 + cr) * q_row + 8 * pc;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                pf[pl][t] = u32x4{0, 0, 0, 0};
                if (in) pf[pl][t] = *reinterpret_cast<const u32x4*>(src + pl * a.DQ);
            }
        }
    };
    const float out_scale = sqrtf(float(HD)) * (1.0f / W_SCALE);   // q carries scale * log2 e: undo the scale
    const bool pairs = !(a.ws & 1);
    float amax = 0.f;
    fetch(0);
    for (int rho0 = 0; rho0 < n_rho; rho0 += 32) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = lane + 64 * t, cr = p / PR, pc = p - cr * PR;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) stage[pl][cr * ST + pc] = pf[pl][t];
        }
        if (rho0 + 32 < n_rho) fetch(rho0 + 32);
        f16x8 qh[KS], ql[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qh[s] = __builtin_bit_cast(f16x8, stage[0][c * ST + 2 * s + h]);
            if constexpr (!PLAIN) ql[s] = __builtin_bit_cast(f16x8, stage[1][c * ST + 2 * s + h]);
        }
        const bool live = rho0 + c < n_rho;
        const size_t my_row = row_of(rho0 + c);
        _Float16* row = Qp + my_row * q_row + HD + axis * a.ws;   // this q row's relative-position columns of the axis
        if constexpr (TAB && PLAIN) row = static_cast<_Float16*>(bias_tab) + my_row * (2 * a.ws) + axis * a.ws;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = mfma_f16(rl[mb][s], qh[s], acc);
                if constexpr (!PLAIN) acc = mfma_f16(rh[mb][s], ql[s], acc);
                acc = mfma_f16(rh[mb][s], qh[s], acc);
            }
            // lane (c, h) holds D[j = 32 mb + 8 g + 4 h + e][rho] in acc[4 g + e]
            if constexpr (TAB && !PLAIN) {   // fp32 table: the same lane swap on four floats, two 16-byte stores (ws % 8 == 0 here)
                float* trow = static_cast<float*>(bias_tab) + my_row * (2 * a.ws) + axis * a.ws;
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    f32x4 mine[2], recv;
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                        for (int e = 0; e < 4; ++e) mine[q2][e] = acc[4 * (2 * p2 + q2) + e] * out_scale;
                    const f32x4 send = h ? mine[0] : mine[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) recv[e] = __shfl_xor(send[e], 32);
                    const int jb = 32 * mb + 16 * p2 + 8 * h;
                    if (live && jb + 8 <= a.ws) {
                        *reinterpret_cast<f32x4*>(trow + jb) = h ? recv : mine[0];
                        *reinterpret_cast<f32x4*>(trow + jb + 4) = h ? mine[1] : recv;
                    }
                }
                continue;
            }
            u32x2 gh[4], gl[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 32 * mb + 8 * g + 4 * h + e < a.ws ? acc[4 * g + e] * out_scale : 0.f;
                amax = pope_amax4(amax, v);
                const float s4 = (v[0] + v[1]) + (v[2] + v[3]);
                if (!(s4 == s4)) amax = INFINITY;   // NaN (fmax drops it)
                f16x4 hi = __builtin_convertvector(v, f16x4), lo = hi;
                if constexpr (!PLAIN) pope_split4(v, hi, lo);
                gh[g] = __builtin_bit_cast(u32x2, hi);
                gl[g] = __builtin_bit_cast(u32x2, lo);
            }
            // lane h = 0 keeps its groups 2 p and takes the partner's (j = 16 p + 0..7), lane h = 1 the groups 2 p + 1 (j = 16 p + 8..15)
            auto put = [&](const u32x2 (&grp)[4], int p2, _Float16* base) {
                const u32x2 send = h ? grp[2 * p2] : grp[2 * p2 + 1];
                const u32x2 recv = {unsigned(__shfl_xor(int(send[0]), 32)), unsigned(__shfl_xor(int(send[1]), 32))};
                const u32x4 out = h ? u32x4{recv[0], recv[1], grp[2 * p2 + 1][0], grp[2 * p2 + 1][1]}
                                    : u32x4{grp[2 * p2][0], grp[2 * p2][1], recv[0], recv[1]};
                const int jb = 32 * mb + 16 * p2 + 8 * h, cnt = a.ws - jb;   // live columns among the lane's eight
                if (!live || cnt <= 0) return;
                _Float16* d = base + jb;
                if (cnt >= 8 && pairs) {
                    *reinterpret_cast<u32x4a4*>(d) = out;
                } else if (pairs) {
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2)
                        if (2 * w2 + 1 < cnt) *reinterpret_cast<unsigned*>(d + 2 * w2) = out[w2];
                } else {   // odd window side: the w axis starts on an odd column
                    const f16x8 o8 = __builtin_bit_cast(f16x8, out);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < cnt) d[e] = o8[e];
                }
            };
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                put(gh, p2, row);
                if constexpr (!PLAIN && !TAB) put(gl, p2, row + a.DQ);
            }
        }
    }
    pope_range_flag(range_flag, POPE_RANGE_QKV, !(amax < POPE_F16_OVERFLOW));
}

// ---- flash attention over the widened operands ----------------------------------------------------------------------
constexpr int KT = 32;

// WAVES x 32 queries per workgroup.  8 (one workgroup per CU) shares each K' / V tile between 256 queries: the global
// blocks, where 16 workgroups walk the same 4096 keys.  4 for the window blocks: a (window, head) is only seven tiles
// long, and two resident workgroups per CU overlap one's prologue / epilogue with the other's tiles (measured at
// ViT-H: +0.7 % on the whole encoder against 8 everywhere, -0.8 % with 4 everywhere).
// PLAIN: single-product f16 arithmetic (precision "f16"): the operands have no lo planes (rows are [DQ] / [DV] halves),
// one MFMA per step, P converted once, output f16 row-major.
template <int NSTEP, int HSTEP, int DVT, int WAVES, bool PLAIN>
struct AttnCfg {
    static constexpr int NT = 64 * WAVES, QB = 32 * WAVES;
    static constexpr int DQ = 16 * NSTEP, HDP = 16 * HSTEP, DV = 32 * DVT;
    static constexpr int KST = DQ + 8, KLST = PLAIN ? 0 : HDP + 8, VST = DV + 8;   // LDS row strides (halves): odd multiples of 16 bytes
    static constexpr int Q_ROW = PLAIN ? DQ : 2 * DQ, K_ROW = PLAIN ? DQ : DQ + HDP, V_ROW = PLAIN ? DV : 2 * DV;   // global rows (halves)
    static constexpr int K_UNITS_ROW = K_ROW / 8, V_UNITS_ROW = V_ROW / 8;   // 16-byte pieces per global row
    static constexpr int K_UNITS = KT * K_UNITS_ROW, V_UNITS = KT * V_UNITS_ROW;
    static constexpr int KP = (K_UNITS + NT - 1) / NT, VP = (V_UNITS + NT - 1) / NT;
    static constexpr int OST = HDP + 4;                                 // epilogue staging row (floats)
    static constexpr size_t STAGE_BYTES = size_t(KT) * (KST + KLST + (PLAIN ? 1 : 2) * VST) * sizeof(_Float16);
    static constexpr size_t EPI_BYTES = size_t(QB) * OST * sizeof(float);
    static constexpr size_t LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
};

// BIAS (the 64 x 64 global blocks): Q' / K' carry q and k only, and the relative-position terms arrive as a bias table
// [G Npad][relh 0..64 | relw 0..64] (`sam_attn_relpos_kernel<.., true>`; fp32, f16 in the plain-f16 mode).  A 32-key tile
// lies inside one key row (kh = kt / 2, kw = 32 (kt & 1) + key), so the score accumulators START from
// relw[q][kw(i)] + relh[q][kh] instead of zero: the lane's 2 x 16 relw values live in registers, relh[q][.] of the
// workgroup's queries in LDS (one broadcast read per tile) — 5 score k-steps per tile instead of 13, K' rows of 80
// columns instead of 208 (profiles/r04/sam_global_bias_ab.txt).
constexpr int BIAS_WS = 64, BIAS_ST = BIAS_WS + 1;
template <int NSTEP, int HSTEP, int DVT, int WAVES, bool PLAIN, bool BIAS>
__global__ __launch_bounds__(64 * WAVES, BIAS && PLAIN && WAVES == 4 ? 3 : 8 / WAVES) void sam_attn_kernel(const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp,
                                                         const _Float16* __restrict__ Vp, const void* __restrict__ bias_tab,
                                                         _Float16* __restrict__ out_pl, AttnGeom a, unsigned* range_flag) {
    using C = AttnCfg<NSTEP, HSTEP, DVT, WAVES, PLAIN>;
    using BT = std::conditional_t<PLAIN, _Float16, float>;   // bias table element
    constexpr int NT = C::NT, QB = C::QB;
    static_assert(!BIAS || NSTEP == HSTEP, "bias mode: no relative-position columns in Q' / K'");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Kh = reinterpret_cast<_Float16*>(smem);
    _Float16* Kl = Kh + KT * C::KST;
    _Float16* Vh = Kl + KT * C::KLST;
    _Float16* Vl = Vh + KT * C::VST;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_qb = (a.Nq + QB - 1) / QB;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);   // the query blocks of one (window, head) share an XCD's L2
    const int grp = logical / n_qb, q0 = (logical - grp * n_qb) * QB;
    const int head = grp % a.heads, wb = grp / a.heads;

    // Q'^T fragments (B operand of S^T = K'.Q'^T): lane (r, h) holds Q'[q = r][16 kg + 8 h + 0..7]; rows past Npad
    // read as zeros (buffer range check) — their waves only keep the barriers company
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(Qp + (size_t)grp * a.Npad * C::Q_ROW), 0, unsigned(a.Npad) * unsigned(C::Q_ROW) * 2u, 0x00020000);
    f16x8 qh[NSTEP], ql[PLAIN ? 1 : NSTEP];
    {
        const unsigned qoff = unsigned(q0 + wave * 32 + r) * unsigned(C::Q_ROW * 2) + unsigned(8 * h) * 2u;
#pragma unroll
        for (int kg = 0; kg < NSTEP; ++kg) {
            qh[kg] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + kg * 32u, 0, 0));
            if constexpr (!PLAIN)
                ql[kg] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + C::DQ * 2u + kg * 32u, 0, 0));
        }
    }

    // K' / V tiles: 32 consecutive rows of the group are one contiguous blob; 16-byte pieces go to the padded LDS rows
    const u32x4* kg_base = reinterpret_cast<const u32x4*>(Kp + (size_t)grp * a.Npad * C::K_ROW);
    const u32x4* vg_base = reinterpret_cast<const u32x4*>(Vp + (size_t)grp * a.Npad * C::V_ROW);
    u32x4 rk[C::KP], rv[C::VP];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < C::KP; ++i) {
            const int u = tid + NT * i;
            if (C::K_UNITS % NT == 0 || u < C::K_UNITS) rk[i] = kg_base[(size_t)kt * C::K_UNITS + u];
        }
#pragma unroll
        for (int i = 0; i < C::VP; ++i) {
            const int u = tid + NT * i;
            if (C::V_UNITS % NT == 0 || u < C::V_UNITS) rv[i] = vg_base[(size_t)kt * C::V_UNITS + u];
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < C::KP; ++i) {
            const int u = tid + NT * i;
            if (C::K_UNITS % NT == 0 || u < C::K_UNITS) {
                const int row = u / C::K_UNITS_ROW, c = u - row * C::K_UNITS_ROW;
                _Float16* dst = c < C::DQ / 8 ? Kh + row * C::KST + c * 8 : Kl + row * C::KLST + (c - C::DQ / 8) * 8;
                *reinterpret_cast<u32x4*>(dst) = rk[i];
            }
        }
#pragma unroll
        for (int i = 0; i < C::VP; ++i) {
            const int u = tid + NT * i;
            if (C::V_UNITS % NT == 0 || u < C::V_UNITS) {
                const int row = u / C::V_UNITS_ROW, c = u - row * C::V_UNITS_ROW;
                _Float16* dst = c < C::DV / 8 ? Vh + row * C::VST + c * 8 : Vl + row * C::VST + (c - C::DV / 8) * 8;
                *reinterpret_cast<u32x4*>(dst) = rv[i];
            }
        }
    };

    // ds_read_b64_tr_b16 addressing of the V^T fragments (A operand of O^T += V^T.P^T), as attention_f16x3.hip: the
    // block of lane l covers keys 4 (l >> 5) + q (+ 16 s, + 8) and value columns 16 ((l >> 4) & 1) + 4 p (+ 32 dt)
    const int tr_off = (4 * h + ((lane & 15) >> 2)) * C::VST + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto vfrag = [&](const _Float16* plane, int s, int dt) {
        const _Float16* p = plane + tr_off + (16 * s) * C::VST + 32 * dt;
        const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
        const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 8 * C::VST));
        return cat(__builtin_bit_cast(f16x4, x), __builtin_bit_cast(f16x4, y));
    };

    f32x16 o[DVT];
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float m_run = -INFINITY;    // running max (log2 domain)
    f32x2 l_run = {0.f, 0.f};   // running sum of the 2^10-scaled probabilities, two partial lanes

    // bias mode: relh[q][0..64) of the workgroup's queries -> LDS behind the stage, the lane's relw values -> registers
    float* Bh = smem + C::STAGE_BYTES / 4;   // (the epilogue's transposition buffer may overlap it: the table is dead by then)
    float bw[BIAS ? 2 : 1][16];
    if constexpr (BIAS) {
        const BT* tab = static_cast<const BT*>(bias_tab) + ((size_t)grp * a.Npad + q0) * (2 * BIAS_WS);
        for (int idx = tid; idx < QB * BIAS_WS; idx += NT) {
            const int qq = idx / BIAS_WS, k = idx - qq * BIAS_WS;
            Bh[qq * BIAS_ST + k] = q0 + qq < a.Npad ? float(tab[(size_t)qq * (2 * BIAS_WS) + k]) : 0.f;
        }
        const bool in = q0 + wave * 32 + r < a.Npad;
        const BT* wrow = tab + (size_t)(wave * 32 + r) * (2 * BIAS_WS) + BIAS_WS + 4 * h;
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) bw[par][4 * g4 + e] = in ? float(wrow[32 * par + 8 * g4 + e]) : 0.f;
    }

    const int nkt = a.Npad / KT;
    load_kv(0);
    auto tile = [&](const int kt, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        if (kt) __syncthreads();   // every wave is done with the previous tile
        store_kv();
        __syncthreads();
        if (kt + 1 < nkt) load_kv(kt + 1);

        // ---- S^T = K'.Q'^T: 3 MFMAs per 16-wide step over q / k proper, 2 over the one-hot columns (no lo plane)
        f32x16 s;
        if constexpr (BIAS) {
            const float bh = Bh[(wave * 32 + r) * BIAS_ST + (kt >> 1)];
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = bw[PAR][i] + bh;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
        }
        const _Float16* kb_h = Kh + r * C::KST + 8 * h;
        const _Float16* kb_l = Kl + r * C::KLST + 8 * h;
#pragma unroll
        for (int kg = 0; kg < NSTEP; ++kg) {
            const f16x8 kh = *reinterpret_cast<const f16x8*>(kb_h + 16 * kg);
            if constexpr (!PLAIN) {
                if (kg < HSTEP) {
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(kb_l + 16 * kg);
                    s = mfma_f16(kl, qh[kg], s);
                }
                s = mfma_f16(kh, ql[kg], s);
            }
            s = mfma_f16(kh, qh[kg], s);
        }
        if (kt + 1 == nkt) {   // keys past the window (rows Nq..Npad of the planes are zeros)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kt * KT + mfma32_row(i, h) >= a.Nq) s[i] = -INFINITY;
        }
        // ---- online softmax in registers (log2 domain; p' = 2^(s - m + 10), the 2^10 cancels in O / l)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s));   // XDL write -> asm VALU read wait states
        float mt = vmax3(s[0], s[1], s[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mt = vmax3(mt, s[i], s[i + 1]);
        mt = __builtin_fmaxf(mt, s[15]);
        mt = __builtin_fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = __builtin_fmaxf(m_run, mt);
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run = l_run * alpha;
#pragma unroll
            for (int dt = 0; dt < DVT; ++dt) o[dt] *= alpha;
        }
        m_run = m_new;
        const float mshift = m_new - 10.0f;
        float ls0 = 0.f, ls1 = 0.f;   // two plain sums (a v_pk_add_f32 beside the other wave's MFMAs costs more than two v_add_f32: attention_f16x3.hip)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            s[i] = __builtin_amdgcn_exp2f(s[i] - mshift);
            s[i + 1] = __builtin_amdgcn_exp2f(s[i + 1] - mshift);
            ls0 += s[i];
            asm volatile("" : "+v"(ls0));
            ls1 += s[i + 1];
            asm volatile("" : "+v"(ls1));
        }
        l_run += f32x2{ls0, ls1};

        // ---- O^T += V^T.P^T: score registers 8 s .. 8 s + 7 are the B fragment of k-step s
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x4 p0, p1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p0[e] = s[8 * st + e];
                p1[e] = s[8 * st + 4 + e];
            }
            if constexpr (PLAIN) {
                const f16x8 ph = cat(__builtin_convertvector(p0, f16x4), __builtin_convertvector(p1, f16x4));
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt) o[dt] = mfma_f16(vfrag(Vh, st, dt), ph, o[dt]);
            } else {
                f16x4 h0, l0, h1, l1;
                pope_split4(p0, h0, l0);
                pope_split4(p1, h1, l1);
                const f16x8 ph = cat(h0, h1), pl = cat(l0, l1);
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt) {
                    const f16x8 vh = vfrag(Vh, st, dt), vl = vfrag(Vl, st, dt);
                    o[dt] = mfma_f16(vl, ph, o[dt]);
                    o[dt] = mfma_f16(vh, pl, o[dt]);
                    o[dt] = mfma_f16(vh, ph, o[dt]);
                }
            }
        }
    };
    if constexpr (BIAS) {   // two tiles per key row: the parity selects the relw registers at compile time
        for (int kt = 0; kt < nkt; kt += 2) {
            tile(kt, std::integral_constant<int, 0>{});
            tile(kt + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) tile(kt, std::integral_constant<int, 0>{});
    }
    __syncthreads();   // the stage is free: reuse it for the O^T transposition

    // normalise, transpose through LDS, un-partition (image_encoder.py:262-285: pad queries are dropped) and write the
    // activation planes of the proj GEMM: [B g g, dim], column head * hd + d
    const float l_half = l_run[0] + l_run[1];
    const float inv = 1.0f / (l_half + __shfl_xor(l_half, 32));
    float* Os = smem + (wave * 32) * C::OST;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (32 * dt + 8 * g4 + 4 * h < C::HDP) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[dt][4 * g4 + e] * inv;
                *reinterpret_cast<f32x4*>(&Os[r * C::OST + 32 * dt + 8 * g4 + 4 * h]) = v;
            }
        }
    __builtin_amdgcn_wave_barrier();
    const int win = wb % (a.nw * a.nw), b = wb / (a.nw * a.nw);
    const int wy = win / a.nw, wx = win - wy * a.nw;
    constexpr int QUADS = C::HDP / 4;   // 16-byte pieces per head row
    f32x2 amax = {0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 32 * QUADS / 64; ++it) {
        const int item = it * 64 + lane, lr = item / QUADS, c4 = (item - lr * QUADS) * 4;
        const int n = q0 + wave * 32 + lr;
        const int y = wy * a.ws + n / a.ws, x = wx * a.ws + n % a.ws;
        if (n < a.Nq && y < a.g && x < a.g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Os[lr * C::OST + c4]);
            pope_amax4x2(amax, v);
            const int col = head * C::HDP + c4;
            const size_t trow = (size_t)b * a.g * a.g + (size_t)y * a.g + x;
            if constexpr (PLAIN) {
                *reinterpret_cast<f16x4*>(out_pl + trow * a.dim + col) = __builtin_convertvector(v * A_SCALE, f16x4);
            } else {
                f16x4 hi, lo;
                pope_split4(v * A_SCALE, hi, lo);
                _Float16* dst = out_pl + trow * 2 * a.dim + (col >> 5) * 64 + (col & 31);
                *reinterpret_cast<f16x4*>(dst) = hi;
                *reinterpret_cast<f16x4*>(dst + 32) = lo;
            }
        }
    }
    pope_range_flag(range_flag, POPE_RANGE_QKV, !(fmaxf(amax[0], amax[1]) * A_SCALE < POPE_F16_OVERFLOW));
}

// ---- precision "f16": LayerNorm over the last dim -> f16 row-major (value * 8), one wave per row (dim % 128 == 0, <= 2048);
// and the plain conversion fp32 -> f16 (value * 8) of the neck's input
template <int NV>
__global__ __launch_bounds__(256) void sam_ln_f16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, _Float16* __restrict__ y, int rows,
                                                         float eps, unsigned* range_flag) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    constexpr int nv = NV, dim = NV * 128;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * dim;
    f32x2 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < nv; ++i) {
        v[i] = *reinterpret_cast<const f32x2*>(xr + i * 128 + lane * 2);
        s += v[i][0] + v[i][1];
    }
    const float mean = wave_sum(s) / float(dim);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < nv; ++i) {
        const float d0 = v[i][0] - mean, d1 = v[i][1] - mean;
        q += d0 * d0 + d1 * d1;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / float(dim) + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < nv; ++i) {
        const int c = i * 128 + lane * 2;
        const f32x2 ww = *reinterpret_cast<const f32x2*>(w + c), bb = *reinterpret_cast<const f32x2*>(b + c);
        f32x2 o;
        o[0] = ((v[i][0] - mean) * rstd * ww[0] + bb[0]) * A_SCALE;
        o[1] = ((v[i][1] - mean) * rstd * ww[1] + bb[1]) * A_SCALE;
        amax = pope_amax2(amax, o);
        *reinterpret_cast<f16x2*>(y + (size_t)row * dim + c) = __builtin_convertvector(o, f16x2);
    }
    pope_range_flag(range_flag, POPE_RANGE_LAYERNORM, !(amax < POPE_F16_OVERFLOW) || !(__builtin_fabsf(mean) + rstd < INFINITY));
}

__global__ __launch_bounds__(256) void sam_to_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, long long n4,
                                                         unsigned* range_flag) {
    float amax = 0.f;
    for (long long id = blockIdx.x * 256ll + threadIdx.x; id < n4; id += 256ll * gridDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * id);
        amax = pope_amax4(amax, v);
        if (!((v[0] + v[1]) + (v[2] + v[3]) == (v[0] + v[1]) + (v[2] + v[3]))) amax = INFINITY;
        *reinterpret_cast<f16x4*>(y + 4 * id) = __builtin_convertvector(v * A_SCALE, f16x4);
    }
    pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax * A_SCALE < POPE_F16_OVERFLOW));
}

// ---- neck LayerNorm2d (common.py:27-43: per pixel over the channels, eps inside the sqrt, a true division) -----------
// in: fp32 [pixels, C] (C % 256 == 0, <= 1024); one wave per pixel, four channels per lane and 256-column group
template <bool TO_BORDERED_PLANES, bool PLAIN = false>
__global__ __launch_bounds__(256) void sam_ln2d_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bvec, void* __restrict__ out, int B, int g, int C,
                                                       float eps, unsigned* range_flag) {
    // TO_BORDERED_PLANES: in = [B g g, C] rows, out = activation planes [B, g + 2, g + 2, C] with a zero border (the
    //   3x3 convolution's operand, conv.hip layout); the wave index walks the BORDERED pixels
    // else: in = fp32 [B, g + 2, g + 2, C] (the convolution's bordered output), out = fp32 NCHW [B, C, g, g]; the wave
    //   index walks the interior pixels
    const int gp = g + 2, nv = C / 256;
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, total = (long long)B * (TO_BORDERED_PLANES ? gp * gp : g * g);
    float amax = 0.f;
    for (long long pix = blockIdx.x * 4ll + (threadIdx.x >> 6); pix < total; pix += waves) {
        int b, y, x;   // interior coordinates
        if (TO_BORDERED_PLANES) {
            b = int(pix / (gp * gp));
            const int rem = int(pix - (long long)b * gp * gp);
            y = rem / gp - 1;
            x = rem % gp - 1;
        } else {
            b = int(pix / (g * g));
            const int rem = int(pix - (long long)b * g * g);
            y = rem / g;
            x = rem % g;
        }
        const bool interior = y >= 0 && y < g && x >= 0 && x < g;
        f32x4 v[4] = {};
        float sum = 0.f;
        const float* src = TO_BORDERED_PLANES ? in + ((size_t)b * g * g + (size_t)y * g + x) * C
                                              : in + ((size_t)b * gp * gp + (size_t)(y + 1) * gp + (x + 1)) * C;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (interior && k < nv) {
                v[k] = *reinterpret_cast<const f32x4*>(src + k * 256 + lane * 4);
                sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
            }
        const float u = wave_sum(sum) / float(C);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (interior && k < nv) {
                v[k] = v[k] - u;
                sq += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
            }
        const float den = sqrtf(wave_sum(sq) / float(C) + eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= nv) continue;
            const int col = k * 256 + lane * 4;
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (interior) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + col), b4 = *reinterpret_cast<const f32x4*>(bvec + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = w4[e] * (v[k][e] / den) + b4[e];
            }
            if (TO_BORDERED_PLANES) {
                amax = pope_amax4(amax, r);
                if (!((r[0] + r[1]) + (r[2] + r[3]) == (r[0] + r[1]) + (r[2] + r[3]))) amax = INFINITY;
                if constexpr (PLAIN) {
                    *reinterpret_cast<f16x4*>(static_cast<_Float16*>(out) + (size_t)pix * C + col) = __builtin_convertvector(r * A_SCALE, f16x4);
                } else {
                    f16x4 hi, lo;
                    pope_split4(r * A_SCALE, hi, lo);
                    _Float16* dst = static_cast<_Float16*>(out) + (size_t)pix * 2 * C + (col >> 5) * 64 + (col & 31);
                    *reinterpret_cast<f16x4*>(dst) = hi;
                    *reinterpret_cast<f16x4*>(dst + 32) = lo;
                }
            } else {
                float* dst = static_cast<float*>(out) + (((size_t)b * C + col) * g + y) * g + x;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(size_t)e * g * g] = r[e];
            }
        }
    }
    if (TO_BORDERED_PLANES) pope_range_flag(range_flag, POPE_RANGE_INPUT, !(amax * A_SCALE < POPE_F16_OVERFLOW));
}

// ---- host ---------------------------------------------------------------------------------------------------------
struct AttnPlan {
    AttnGeom geom;
    int nstep;      // DQ / 16
    bool bias;      // relative-position terms as a bias table (sam_attn_kernel<.., BIAS>), not as columns of Q' / K'
    size_t qp, kp, vp, tab;   // bytes
};

// the smallest instantiated score depth that holds hd + 2 ws columns
bool plan_attention(int B, int g, int ws, int heads, int hd, AttnPlan& p) {
    AttnGeom& a = p.geom;
    a.B = B; a.g = g; a.ws = ws; a.nw = (g + ws - 1) / ws;
    a.heads = heads; a.hd = hd; a.dim = heads * hd;
    a.Nq = ws * ws; a.Npad = (a.Nq + KT - 1) / KT * KT;
    a.HDP = hd; a.DV = hd == 80 ? 96 : 64;
    const int need = hd + 2 * ws;
    static const int depths80[] = {7, 13}, depths64[] = {6, 12};
    const int* d = hd == 80 ? depths80 : depths64;
    p.bias = ws == BIAS_WS && a.nw == 1;   // the 64 x 64 global blocks
    p.nstep = p.bias ? hd / 16 : 16 * d[0] >= need ? d[0] : (16 * d[1] >= need ? d[1] : 0);
    if (!p.nstep) return false;
    a.DQ = 16 * p.nstep;
    const size_t G = size_t(B) * a.nw * a.nw * heads;
    p.tab = p.bias ? G * a.Npad * 2 * ws * sizeof(float) : 0;
    p.qp = G * a.Npad * 2 * a.DQ * sizeof(_Float16);
    p.kp = G * a.Npad * (a.DQ + a.HDP) * sizeof(_Float16);
    p.vp = G * a.Npad * 2 * a.DV * sizeof(_Float16);
    return true;
}

template <int NSTEP, int HSTEP, int DVT, int WAVES, bool PLAIN, bool BIAS = false>
int launch_attn(const AttnPlan& p, const _Float16* Qp, const _Float16* Kp, const _Float16* Vp, const void* bias_tab, _Float16* out,
                unsigned* flag, hipStream_t stream) {
    using C = AttnCfg<NSTEP, HSTEP, DVT, WAVES, PLAIN>;
    constexpr int NT = C::NT, QB = C::QB;
    constexpr size_t bias_lds = C::STAGE_BYTES + size_t(QB) * BIAS_ST * sizeof(float);
    constexpr size_t lds = BIAS && bias_lds > C::LDS_BYTES ? bias_lds : C::LDS_BYTES;
    static pope_dev_mask done{0};
    auto kern = sam_attn_kernel<NSTEP, HSTEP, DVT, WAVES, PLAIN, BIAS>;
    if (!pope_opt_in_lds(kern, lds, done)) return POPE_ERR_LAUNCH;
    const AttnGeom& a = p.geom;
    if (BIAS && (a.ws != BIAS_WS || a.nw != 1 || (a.Npad / KT) % 2 || !bias_tab)) return POPE_ERR_ARG;
    const long long blocks = (long long)a.B * a.nw * a.nw * a.heads * ((a.Nq + QB - 1) / QB);
    if (blocks <= 0 || blocks > 0x7fffffffll) return POPE_ERR_ARG;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, stream, Qp, Kp, Vp, bias_tab, out, a, flag);
    return pope_check_launch();
}

}  // namespace

// y = LayerNorm(x) * w + b as f16 row-major (value * 8): the A operand of the plain-f16 GEMMs (POPE_PREC_F16)
int pope_launch_layernorm_f16(const float* x, const float* w, const float* b, void* y_f16, int rows, int dim, float eps, unsigned* flag,
                              hipStream_t stream) {
    if (!x || !w || !b || !y_f16 || rows <= 0) return POPE_ERR_ARG;
#define POPE_SAM_LN(NV)                                                                                                       \
    case NV:                                                                                                                  \
        hipLaunchKernelGGL(sam_ln_f16_kernel<NV>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, w, b, static_cast<_Float16*>(y_f16), \
                           rows, eps, flag);                                                                                  \
        break;
    switch (dim % 128 ? 0 : dim / 128) {
        POPE_SAM_LN(2) POPE_SAM_LN(3) POPE_SAM_LN(4) POPE_SAM_LN(5) POPE_SAM_LN(6) POPE_SAM_LN(8) POPE_SAM_LN(10) POPE_SAM_LN(12) POPE_SAM_LN(16)
        default: return POPE_ERR_ARG;
    }
#undef POPE_SAM_LN
    return pope_check_launch();
}

size_t pope_sam_encoder_workspace(const SamEncParams& q) {
    if (q.B <= 0 || q.img <= 0 || q.patch <= 0 || q.img % q.patch || q.heads <= 0 || q.dim % q.heads) return 0;
    const int g = q.img / q.patch, hd = q.dim / q.heads;
    const size_t rows = size_t(q.B) * g * g;
    const size_t kp = size_t(3) * q.patch * q.patch;
    size_t big = rows * 4 * q.dim * 4;                      // attention output planes [rows, dim] (room for 4 dim: mlp_ratio 4)
    if (rows * q.hidden * 4 > big) big = rows * q.hidden * 4;   // fc1 output planes
    if (rows * kp * 4 > big) big = rows * kp * 4;               // im2col planes
    size_t ops = 0;   // one operand set per geometry (window, global): their constant parts are written once per forward pass
    for (int pass = 0; pass < 2; ++pass) {
        AttnPlan p;
        const int ws = pass ? g : (q.window > 0 ? q.window : g);
        if (!plan_attention(q.B, g, ws, q.heads, hd, p)) return 0;
        ops += align256(p.qp) + align256(p.kp) + align256(p.vp) + align256(p.tab) + align256(rows * sizeof(int));
    }
    const size_t gp = size_t(g) + 2;
    return align256(rows * q.dim * 4) /* x */ + align256(rows * q.dim * 4) /* xn planes */ + align256(big) + ops +
           align256(rows * q.out_chans * 4) /* neck 1x1 */ + 2 * align256(size_t(q.B) * gp * gp * q.out_chans * 4);
}

int pope_launch_sam_encoder(const SamEncParams& q, hipStream_t stream) {
    if (!q.image || !q.out || !q.ws || !q.blocks || !q.patch_wp || !q.patch_b || !q.ones || !q.neck0_wp || !q.neck2_wp ||
        !q.neck1_w || !q.neck1_b || !q.neck3_w || !q.neck3_b)
        return POPE_ERR_ARG;
    if (q.B <= 0 || q.depth <= 0 || q.patch <= 0 || (q.patch & 7) || q.img % q.patch || q.heads <= 0 || q.dim % q.heads)
        return POPE_ERR_ARG;
    const int g = q.img / q.patch, hd = q.dim / q.heads, dim = q.dim, hidden = q.hidden, oc = q.out_chans;
    if ((hd != 64 && hd != 80) || (dim & 127) || dim > 2048 || (hidden & 31) || (oc & 255) || oc > 1024 || q.window < 0) return POPE_ERR_ARG;
    const int kp = 3 * q.patch * q.patch;
    if (kp & 31) return POPE_ERR_ARG;
    const size_t need = pope_sam_encoder_workspace(q);
    if (!need) return POPE_ERR_ARG;
    if (q.ws_bytes < need) return POPE_ERR_WORKSPACE;
    const int rows = q.B * g * g;
    const size_t gp = size_t(g) + 2, brows = size_t(q.B) * gp * gp;
    if (size_t(rows + 256) * (hidden > 3 * dim ? hidden : 3 * dim) * 4 >= (1ull << 32) - 512 || brows * oc * 4 >= (1ull << 32) - 512)
        return POPE_ERR_ARG;   // 32-bit buffer offsets in the GEMMs: the caller splits larger batches

    AttnPlan plan_w, plan_g;
    if (!plan_attention(q.B, g, q.window > 0 ? q.window : g, q.heads, hd, plan_w) || !plan_attention(q.B, g, g, q.heads, hd, plan_g))
        return POPE_ERR_ARG;

    char* base = static_cast<char*>(q.ws);
    auto take = [&](size_t bytes) { char* p = base; base += align256(bytes); return p; };
    float* x = reinterpret_cast<float*>(take(size_t(rows) * dim * 4));
    void* xn_pl = take(size_t(rows) * dim * 4);
    size_t big_bytes = size_t(rows) * 4 * dim * 4;
    if (size_t(rows) * hidden * 4 > big_bytes) big_bytes = size_t(rows) * hidden * 4;
    if (size_t(rows) * kp * 4 > big_bytes) big_bytes = size_t(rows) * kp * 4;
    char* big = take(big_bytes);
    void* att_pl = big;   // attention output (the proj GEMM's operand); fc1's output reuses the buffer
    void* hid_pl = big;
    struct OpSet { _Float16 *q, *k, *v; void* tab; int* map; };
    OpSet ops_w, ops_g;
    for (auto pr : {std::make_pair(&plan_w, &ops_w), std::make_pair(&plan_g, &ops_g)}) {
        pr.second->q = reinterpret_cast<_Float16*>(take(pr.first->qp));
        pr.second->k = reinterpret_cast<_Float16*>(take(pr.first->kp));
        pr.second->v = reinterpret_cast<_Float16*>(take(pr.first->vp));
        pr.second->tab = take(pr.first->tab);
        pr.second->map = reinterpret_cast<int*>(take(size_t(rows) * sizeof(int)));
    }
    float* t1 = reinterpret_cast<float*>(take(size_t(rows) * oc * 4));
    void* t1_pl = take(brows * oc * 4);
    float* t2 = reinterpret_cast<float*>(take(brows * oc * 4));

    // LayerNorm eps of the blocks (build_sam.py:71 passes 1e-6; the constructor's default norm_layer has 1e-5) and of the
    // neck's LayerNorm2d (common.py:28: 1e-6)
    const float eps = q.block_eps > 0.f ? q.block_eps : 1e-6f, neck_eps = q.neck_eps > 0.f ? q.neck_eps : 1e-6f;
    unsigned* flag = q.range_flag;
    // precision "f16" (POPE_PREC_F16): every operand is a plain f16 tensor (activations * 8, weights * 256), one MFMA per
    // product, fp32 accumulation, fp32 residual stream / softmax / LayerNorm statistics — BASELINE config 5's dtype
    if (q.precision == POPE_PREC_F32_MFMA) return pope_launch_sam_encoder_f32mfma(q, stream);   // sam_f32.hip: `*_wp` are fp32 matrices
    if (q.precision != POPE_PREC_F16X3 && q.precision != POPE_PREC_F16) return POPE_ERR_ARG;
    const bool plain = q.precision == POPE_PREC_F16;
    if (plain && ((dim & 63) || (hidden & 63) || (kp & 63))) return POPE_ERR_ARG;
    int rc;
#define POPE_TRY(call) do { if ((rc = (call))) return rc; } while (0)
    auto gemm = [&](const void* a_pl, const void* w_pl, const float* bias, float* Cf, void* c_pl, int N, int K, int epi,
                    const float* gamma, const float* res, int res_mod) {
        GemmParams gp_ = {};
        gp_.range_flag = flag;
        gp_.range_bit = epi == EPI_BIAS_GELU ? POPE_RANGE_GELU : POPE_RANGE_QKV;
        gp_.a_pl = a_pl; gp_.w_pl = w_pl; gp_.bias = bias; gp_.C = Cf; gp_.c_pl = c_pl;
        const int Kc = plain ? K / 2 : K;   // plain: columns are counted in pairs (GemmParams::plain)
        gp_.lda = Kc; gp_.ldw = Kc; gp_.ldc = plain && c_pl ? N / 2 : N; gp_.M = rows; gp_.N = N; gp_.K = Kc;
        gp_.epilogue = epi; gp_.gamma = gamma; gp_.res = res; gp_.ldres = N; gp_.res_mod = res_mod;
        gp_.plain = plain;
        return pope_launch_gemm_nt_f16x3_planes(gp_, stream);
    };
    auto layernorm = [&](const float* w, const float* b) -> int {   // LN(x) -> xn_pl as this precision's GEMM operand
        if (!plain) return pope_launch_layernorm_planes(x, dim, w, b, xn_pl, rows, dim, eps, flag, stream);
        return pope_launch_layernorm_f16(x, w, b, xn_pl, rows, dim, eps, flag, stream);
    };

    // patch embed + absolute position table (image_encoder.py:108-110): x = conv(img) + bias + pos[token]
    {
        const long long total = (long long)rows * (kp / 8);
        if (plain)
            hipLaunchKernelGGL(sam_im2col_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, q.image,
                               reinterpret_cast<_Float16*>(big), q.B, q.img, q.patch, flag);
        else
            hipLaunchKernelGGL(sam_im2col_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, q.image,
                               reinterpret_cast<_Float16*>(big), q.B, q.img, q.patch, flag);
        POPE_TRY(pope_check_launch());
        if (q.pos) POPE_TRY(gemm(big, q.patch_wp, q.patch_b, x, nullptr, dim, kp, EPI_BIAS_LS_RES, q.ones, q.pos, g * g));
        else POPE_TRY(gemm(big, q.patch_wp, q.patch_b, x, nullptr, dim, kp, EPI_BIAS, nullptr, nullptr, 0));
    }
    // The QKV GEMM's epilogue writes q * scale, k, v into the attention operand rows (EPI_SAM_QKV).  What does not depend on
    // the block — zero rows and columns, K's one-hot columns, the window partition's row map — is written here, once per
    // forward pass and geometry.
    {
        bool use[2] = {false, false};
        for (int i = 0; i < q.depth; ++i) use[(q.blocks[i].global || q.window <= 0) ? 1 : 0] = true;
        for (int s2 = 0; s2 < 2; ++s2) {
            if (!use[s2]) continue;
            const AttnPlan& p = s2 ? plan_g : plan_w;
            const OpSet& os = s2 ? ops_g : ops_w;
            if (hipMemsetAsync(os.q, 0, p.qp, stream) != hipSuccess || hipMemsetAsync(os.k, 0, p.kp, stream) != hipSuccess ||
                hipMemsetAsync(os.v, 0, p.vp, stream) != hipSuccess)
                return POPE_ERR_LAUNCH;
            const AttnGeom& a = p.geom;
            const int k_row = plain ? a.DQ : a.DQ + a.HDP;
            if (!p.bias) {   // (bias mode: K' has no one-hot columns)
                hipLaunchKernelGGL(sam_onehot_kernel, dim3(grid_for((long long)a.B * a.nw * a.nw * a.heads * a.Nq)), dim3(256), 0, stream,
                                   os.k, a, k_row);
                POPE_TRY(pope_check_launch());
            }
            hipLaunchKernelGGL(sam_rowmap_kernel, dim3(grid_for(rows)), dim3(256), 0, stream, os.map, a);
            POPE_TRY(pope_check_launch());
        }
    }
    for (int i = 0; i < q.depth; ++i) {
        const SamBlockParams& k = q.blocks[i];
        if (!k.norm1_w || !k.norm1_b || !k.qkv_wp || !k.qkv_b || !k.proj_wp || !k.proj_b || !k.norm2_w || !k.norm2_b || !k.fc1_wp ||
            !k.fc1_b || !k.fc2_wp || !k.fc2_b || !k.rel_h || !k.rel_w)
            return POPE_ERR_ARG;
        const bool glob = k.global || q.window <= 0;
        const AttnPlan& p = glob ? plan_g : plan_w;
        const OpSet& os = glob ? ops_g : ops_w;
        const AttnGeom& a = p.geom;
        // x = x + attn(norm1(x))                                         image_encoder.py:166-179
        POPE_TRY(layernorm(k.norm1_w, k.norm1_b));
        _Float16 *Qp = os.q, *Kp = os.k, *Vp = os.v;
        {
            // QKV projection written straight into the operand rows (window partition = os.map, q * scale * log2 e) ...
            GemmParams gq = {};
            gq.range_flag = flag; gq.range_bit = POPE_RANGE_QKV;
            gq.a_pl = xn_pl; gq.w_pl = k.qkv_wp; gq.bias = k.qkv_b; gq.c_pl = Qp;
            const int Kc = plain ? dim / 2 : dim;
            gq.lda = Kc; gq.ldw = Kc; gq.K = Kc; gq.ldc = 32; gq.M = rows; gq.N = 3 * dim;
            gq.epilogue = EPI_SAM_QKV; gq.plain = plain;
            gq.sam_q = Qp; gq.sam_k = Kp; gq.sam_v = Vp; gq.sam_rowmap = os.map;
            if (p.qp >= (1ull << 32) || p.kp >= (1ull << 32) || p.vp >= (1ull << 32)) return POPE_ERR_ARG;   // 32-bit operand offsets
            gq.sam_bytes[0] = unsigned(p.qp); gq.sam_bytes[1] = unsigned(p.kp); gq.sam_bytes[2] = unsigned(p.vp);
            gq.sam_hd = hd; gq.sam_dim = dim; gq.sam_npad = a.Npad; gq.sam_dq = a.DQ; gq.sam_dv = a.DV;
            gq.sam_qscale = 1.0f / sqrtf(float(hd)) * L2E;
            POPE_TRY(pope_launch_planes16(gq, stream));
            if (a.nw * a.ws > a.g) {   // ... the rows of the edge windows' zero-padded tokens from the bias ...
                const long long total = (long long)a.B * (a.nw * a.ws * a.nw * a.ws - a.g * a.g) * a.heads * 3 * (hd / 8);
                if (plain)
                    hipLaunchKernelGGL(sam_pad_tokens_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, k.qkv_b, Qp, Kp, Vp, a, flag);
                else
                    hipLaunchKernelGGL(sam_pad_tokens_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, k.qkv_b, Qp, Kp, Vp, a, flag);
                POPE_TRY(pope_check_launch());
            }
            // ... and the relative-position columns of Q' from the Q' rows
            int hpg = 256 / a.ws < 1 ? 1 : (256 / a.ws > a.heads ? a.heads : 256 / a.ws);   // heads per wave: about 256 q rows
            while (a.heads % hpg) --hpg;
            const long long n_tasks = (long long)a.B * a.nw * a.nw * 2 * a.ws * (a.heads / hpg);
            if (a.ws > 64 || n_tasks > 0x7ffffff0ll) return POPE_ERR_ARG;
            const dim3 rgrid((unsigned)((n_tasks + 3) / 4));
            const unsigned ws_magic = unsigned(((1ull << 32) + a.ws - 1) / a.ws);   // rho / ws = umulhi(rho, magic) for rho < 2^16
            // both axes of one window batch on one XCD while its Q' rows fit that L2 comfortably (the 28 tasks of a 14 x 14 window: yes;
            // a 64 x 64 global block: no — measured slower, profiles/r04/sam_relpos_lab.txt)
            const int by_xcd = size_t(a.Nq) * a.heads * (plain ? a.DQ : 2 * a.DQ) * 2 <= (1u << 20);
#define POPE_SAM_RELPOS(HD, MB, PL, TAB)                                                                                               \
    hipLaunchKernelGGL((sam_attn_relpos_kernel<HD, MB, PL, TAB>), rgrid, dim3(256), 0, stream, k.rel_h, k.rel_w, Qp, os.tab, a, hpg,       \
                       int(n_tasks), ws_magic, by_xcd, flag)
#define POPE_SAM_RELPOS_MB(HD, PL)                               \
    do {                                                         \
        if (p.bias) POPE_SAM_RELPOS(HD, 2, PL, true);            \
        else if (a.ws > 32) POPE_SAM_RELPOS(HD, 2, PL, false);   \
        else POPE_SAM_RELPOS(HD, 1, PL, false);                  \
    } while (0)
            if (hd == 80) { if (plain) POPE_SAM_RELPOS_MB(80, true); else POPE_SAM_RELPOS_MB(80, false); }
            else { if (plain) POPE_SAM_RELPOS_MB(64, true); else POPE_SAM_RELPOS_MB(64, false); }
#undef POPE_SAM_RELPOS_MB
#undef POPE_SAM_RELPOS
            POPE_TRY(pope_check_launch());
        }
        _Float16* att = static_cast<_Float16*>(att_pl);
        const bool narrow = a.Nq <= 1024;   // window blocks: 4-wave workgroups; global blocks: 8
#define POPE_SAM_ATTN_W(NS, HS, DV, W)                                                            \
    (plain ? launch_attn<NS, HS, DV, W, true>(p, Qp, Kp, Vp, nullptr, att, flag, stream)          \
           : launch_attn<NS, HS, DV, W, false>(p, Qp, Kp, Vp, nullptr, att, flag, stream))
#define POPE_SAM_ATTN(NS, HS, DV) (narrow ? POPE_SAM_ATTN_W(NS, HS, DV, 4) : POPE_SAM_ATTN_W(NS, HS, DV, 8))
        // bias mode, plain f16: 4-wave workgroups whose bias table shares the epilogue's LDS (45 KB: three per CU) at <= 168 VGPRs =
        // three waves per SIMD instead of two: encoder 7.275 -> 7.20 ms per image (profiles/r04/sam_global_bias_ab.txt)
#define POPE_SAM_ATTN_BIAS(HS, DV)                                                                \
    (plain ? launch_attn<HS, HS, DV, 4, true, true>(p, Qp, Kp, Vp, os.tab, att, flag, stream)     \
           : launch_attn<HS, HS, DV, 8, false, true>(p, Qp, Kp, Vp, os.tab, att, flag, stream))
        if (p.bias) POPE_TRY(hd == 80 ? POPE_SAM_ATTN_BIAS(5, 3) : POPE_SAM_ATTN_BIAS(4, 2));
        else if (hd == 80) POPE_TRY(p.nstep == 7 ? POPE_SAM_ATTN(7, 5, 3) : POPE_SAM_ATTN(13, 5, 3));
        else POPE_TRY(p.nstep == 6 ? POPE_SAM_ATTN(6, 4, 2) : POPE_SAM_ATTN(12, 4, 2));
#undef POPE_SAM_ATTN_BIAS
#undef POPE_SAM_ATTN
#undef POPE_SAM_ATTN_W
        POPE_TRY(gemm(att_pl, k.proj_wp, k.proj_b, x, nullptr, dim, dim, EPI_BIAS_LS_RES, q.ones, x, 0));
        // x = x + mlp(norm2(x))                                          image_encoder.py:181; common.py:13-25
        POPE_TRY(layernorm(k.norm2_w, k.norm2_b));
        POPE_TRY(gemm(xn_pl, k.fc1_wp, k.fc1_b, nullptr, hid_pl, hidden, dim, EPI_BIAS_GELU, nullptr, nullptr, 0));
        POPE_TRY(gemm(hid_pl, k.fc2_wp, k.fc2_b, x, nullptr, dim, hidden, EPI_BIAS_LS_RES, q.ones, x, 0));
        for (int t = 0; t < q.n_taps; ++t)
            if (q.tap_blocks[t] == i && q.tap_out[t] &&
                hipMemcpyAsync(q.tap_out[t], x, size_t(rows) * dim * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return POPE_ERR_LAUNCH;
    }
    // neck (image_encoder.py:89-105): 1x1 conv (no bias) -> LayerNorm2d -> 3x3 conv pad 1 (no bias) -> LayerNorm2d
    if (plain) {
        const long long n4 = (long long)rows * dim / 4;
        hipLaunchKernelGGL(sam_to_f16_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, x, static_cast<_Float16*>(xn_pl), n4, flag);
        POPE_TRY(pope_check_launch());
    } else {
        POPE_TRY(pope_launch_split_planes(x, xn_pl, rows, dim, A_SCALE, flag, stream));
    }
    POPE_TRY(gemm(xn_pl, q.neck0_wp, nullptr, t1, nullptr, oc, dim, EPI_BIAS, nullptr, nullptr, 0));
    if (plain)
        hipLaunchKernelGGL((sam_ln2d_kernel<true, true>), dim3(grid_for((long long)brows, 4)), dim3(256), 0, stream, t1, q.neck1_w,
                           q.neck1_b, t1_pl, q.B, g, oc, neck_eps, flag);
    else
        hipLaunchKernelGGL((sam_ln2d_kernel<true, false>), dim3(grid_for((long long)brows, 4)), dim3(256), 0, stream, t1, q.neck1_w,
                           q.neck1_b, t1_pl, q.B, g, oc, neck_eps, flag);
    POPE_TRY(pope_check_launch());
    {
        GemmParams c = {};
        const int Wp = g + 2;
        const size_t shift = size_t(Wp) + 1;   // output row R is pixel R + Wp + 1 (conv.hip)
        c.a_pl = t1_pl; c.w_pl = q.neck2_wp; c.bias = nullptr;
        const int occ = plain ? oc / 2 : oc;   // plain: column pairs
        c.lda = occ; c.ldw = 9 * occ; c.ldc = oc;
        c.M = int(brows - (2 * size_t(Wp) + 2)); c.N = oc; c.K = 9 * occ;
        c.plain = plain;
        c.epilogue = EPI_CONV; c.act_slope = 1.0f;   // identity
        c.C = t2 + shift * oc;
        c.conv_cch = occ / 32; c.conv_wp = Wp;
        c.range_flag = flag; c.range_bit = POPE_RANGE_INPUT;
        c.nbatch = 1;
        POPE_TRY(pope_launch_planes16(c, stream));
    }
    hipLaunchKernelGGL((sam_ln2d_kernel<false, false>), dim3(grid_for((long long)rows, 4)), dim3(256), 0, stream, t2, q.neck3_w, q.neck3_b, q.out,
                       q.B, g, oc, neck_eps, nullptr);
    POPE_TRY(pope_check_launch());
#undef POPE_TRY
    return POPE_OK;
}
