// Flash-style multi-head attention on the f16 matrix cores with error-compensated operands
// ("f16x3": x = hi + lo, three v_mfma_f32_32x32x16_f16 per product block, fp32 accumulate).
// Same contract as attention_f32.hip (attention.py:49-62: qkv[B,N,3,H,64] fp32 -> out[B,N,H*64]
// fp32), same orientation: S^T = K.Q^T (16 keys of one query per lane), O^T += V^T.P^T with the
// score registers converted in place into the B operand of the second product.
//
// Why: on gfx950 the f32 MFMA runs on the VALU lanes (157 TF/s, softmax VALU work is paid in full
// on top); the f16 MFMA has 16x the rate on a separate pipe, so 3 MFMAs per product are 5.3x faster
// and the softmax overlaps.  Accuracy: hi+lo carries 22 significand bits; Q is pre-scaled to the
// log2 domain, P is computed as 2^(s - m + 10) (the 2^10 cancels in O / l) so that every probability
// down to 1e-4 keeps a normal-range lo half; measured error of the output is at the level of the
// fp32 chain (tests/test_gpu_ops.py).  Range contract: |q|,|k|,|v| < 65504.
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int HD = 64, KT = 64;
constexpr int WAVES = 8;          // 8 waves x 32 queries share each K/V tile: half the L2 -> LDS traffic and half the
constexpr int QB = 32 * WAVES;    // per-wave K/V split work of a 4-wave block (the kernel is L2-bandwidth sensitive)
constexpr int NT = 64 * WAVES;
constexpr int KST = 72;   // K plane row stride (halves): 144 B = 9 x 16 B (odd) -> ds_read_b128 rows conflict-free
constexpr int VST = 96;   // V plane row stride (halves): 192 B -> the 4 rows of a ds_read_b64_tr_b16 block hit disjoint banks
// + 32 halves: the lo plane starts 16 banks after the hi plane, so the eight lanes that copy one 128-byte planes chunk
// (4 hi pieces + 4 lo pieces) into LDS hit 32 different store banks (without it: a 2-way conflict on every ds_write_b128)
constexpr int K_PLANE = KT * KST + 32, V_PLANE = KT * VST + 32;
constexpr size_t X3_ATTN_STAGE_BYTES = size_t(2) * (K_PLANE + V_PLANE) * sizeof(_Float16);  // 43 008 B
constexpr int OST = 68;   // epilogue staging row (floats)
constexpr size_t X3_ATTN_EPI_BYTES = size_t(32) * 8 * OST * sizeof(float);  // O^T transpose staging, 32 rows per wave


__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split4(f32x4 v, f16x4& hi, f16x4& lo) { pope_split4(v, hi, lo); }  // common.h
__device__ __forceinline__ f16x8 cat(f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// OUT_PLANES: write the output as f16 hi/lo activation planes (pope_hip.h layout, scale 8) for the
// f16x3 proj GEMM instead of fp32.
// (POPE_PREC_F16's single-product attention lives in attention_f16.hip since round 4.)
template <bool OUT_PLANES>
__global__ __launch_bounds__(NT, 2) void attn_f16x3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             int N, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Kh = reinterpret_cast<_Float16*>(smem);
    _Float16* Kl = Kh + K_PLANE;
    _Float16* Vh = Kl + K_PLANE;
    _Float16* Vl = Vh + V_PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_qb = (N + QB - 1) / QB;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);  // query blocks of one (image, head) share an XCD's L2
    const int bh = logical / n_qb, head = bh % heads, b = bh / heads, q0 = (logical - bh * n_qb) * QB;
    const int D = heads * HD, rs = 3 * D;
    const float* base = qkv + size_t(b) * N * rs;
    const int koff = D + head * HD;

    // Q^T fragments (B operand of S^T = K.Q^T): lane (r,h) holds Q[q = r][d = 16kg + 8h + j], scaled by
    // head_dim^-0.5 * log2(e) so that the scores leave the MFMA in the log2 domain.
    constexpr float QSCALE = 0.125f * 1.44269504088896340736f;
    f16x8 qh[4], ql[4];
    {
        const int qrow = q0 + wave * 32 + r;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (qrow < N) {
                const float* p = base + size_t(qrow) * rs + head * HD + 16 * kg + 8 * h;
                v0 = *reinterpret_cast<const f32x4*>(p);
                v1 = *reinterpret_cast<const f32x4*>(p + 4);
            }
            f16x4 h0, l0, h1, l1;
            split4(v0 * QSCALE, h0, l0);
            split4(v1 * QSCALE, h1, l1);
            qh[kg] = cat(h0, h1);
            ql[kg] = cat(l0, l1);
        }
    }

    // K/V staging: bounds-checked buffer loads (keys >= N read as zeros), split into hi/lo planes on
    // the way into LDS (row-major [key][d]; V is consumed through the transposing LDS read).
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, unsigned(N) * unsigned(rs) * 4u, 0x00020000);
    constexpr int RPP = NT / 16, NP = KT / RPP;  // staging rows per pass, passes per 64-key tile
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    unsigned kvoff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) kvoff[i] = (unsigned(srow + RPP * i) * unsigned(rs) + scol + koff) * 4u;
    const unsigned tile_bytes = unsigned(KT) * unsigned(rs) * 4u, v_delta = unsigned(D) * 4u;
    f32x4 rk[NP], rv[NP];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff[i], kt * tile_bytes, 0));
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff[i] + v_delta, kt * tile_bytes, 0));
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            f16x4 hi, lo;
            split4(rk[i], hi, lo);
            *reinterpret_cast<f16x4*>(Kh + (srow + RPP * i) * KST + scol) = hi;
            *reinterpret_cast<f16x4*>(Kl + (srow + RPP * i) * KST + scol) = lo;
            split4(rv[i], hi, lo);
            *reinterpret_cast<f16x4*>(Vh + (srow + RPP * i) * VST + scol) = hi;
            *reinterpret_cast<f16x4*>(Vl + (srow + RPP * i) * VST + scol) = lo;
        }
    };

    // ds_read_b64_tr_b16 addressing for the V^T fragments (A operand of O^T += V^T.P^T): within a
    // 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3 of a 4-key x 16-d block and lane i
    // receives column i (its d) of the 4 keys.  Block of lane l: keys 4*(l>>5) + q (+16s +8 +32u),
    // d columns 16*((l>>4)&1) + 4p (+32dt).
    const int tr_off = (4 * h + ((lane & 15) >> 2)) * VST + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto vfrag = [&](const _Float16* plane, int u, int s, int dt) {
        const _Float16* p = plane + tr_off + (32 * u + 16 * s) * VST + 32 * dt;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
        const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 8 * VST));
        return cat(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, c));
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY;   // running max (log2 domain)
    f32x2 l_run = {0.f, 0.f};  // running sum of 2^10-scaled probabilities, two partial lanes

    auto tile = [&](int kt, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        if (kt) __syncthreads();  // every wave is done with the previous K/V stage
        store_kv();
        __syncthreads();
        if constexpr (!LAST) load_kv(kt + 1);

        // ---- S^T = K . Q^T, two 32-key sub-tiles, 3 MFMAs per 16-wide d step --------------------
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        const _Float16* kb_h = Kh + r * KST + 8 * h;
        const _Float16* kb_l = Kl + r * KST + 8 * h;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const f16x8 k0h = *reinterpret_cast<const f16x8*>(kb_h + 16 * kg);
            const f16x8 k1h = *reinterpret_cast<const f16x8*>(kb_h + 32 * KST + 16 * kg);
            const f16x8 k0l = *reinterpret_cast<const f16x8*>(kb_l + 16 * kg);
            const f16x8 k1l = *reinterpret_cast<const f16x8*>(kb_l + 32 * KST + 16 * kg);
            s0 = mfma_f16(k0l, qh[kg], s0);
            s1 = mfma_f16(k1l, qh[kg], s1);
            s0 = mfma_f16(k0h, ql[kg], s0);
            s1 = mfma_f16(k1h, ql[kg], s1);
            s0 = mfma_f16(k0h, qh[kg], s0);
            s1 = mfma_f16(k1h, qh[kg], s1);
        }
        if constexpr (LAST) {  // mask the padded keys of the last tile
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kt * KT + mfma32_row(i, h);
                if (key >= N) s0[i] = -INFINITY;
                if (key + 32 >= N) s1[i] = -INFINITY;
            }
        }
        // ---- online softmax in registers (log2 domain; p' = 2^(s - m + 10)) ------------------------
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s0), "+v"(s1));  // XDL write -> asm VALU read wait states
        float mt = vmax3(s0[0], s1[0], s0[1]);
#pragma unroll
        for (int i = 1; i < 15; ++i) mt = vmax3(mt, s1[i], s0[i + 1]);
        mt = vmax3(mt, s1[15], s1[15]);
        mt = __builtin_fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = __builtin_fmaxf(m_run, mt);
        if (__any(m_new > m_run)) {  // rescale only when some row's max moved (exact: alpha == 1 otherwise)
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run = l_run * alpha;
            o0 *= alpha;
            o1 *= alpha;
        }
        m_run = m_new;
        const float mshift = m_new - 10.0f;
        f32x2 ls = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            s0[i] = __builtin_amdgcn_exp2f(s0[i] - mshift);
            s0[i + 1] = __builtin_amdgcn_exp2f(s0[i + 1] - mshift);
            s1[i] = __builtin_amdgcn_exp2f(s1[i] - mshift);
            s1[i + 1] = __builtin_amdgcn_exp2f(s1[i + 1] - mshift);
            ls += f32x2{s0[i], s0[i + 1]} + f32x2{s1[i], s1[i + 1]};
        }
        l_run += ls;

        // ---- O^T += V^T . P^T: score registers 8s..8s+7 of sub-tile u are the B fragment of k-step (u,s)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 p0, p1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p0[e] = (u ? s1 : s0)[8 * s + e];
                    p1[e] = (u ? s1 : s0)[8 * s + 4 + e];
                }
                f16x4 h0, l0, h1, l1;
                split4(p0, h0, l0);
                split4(p1, h1, l1);
                const f16x8 ph = cat(h0, h1), pl = cat(l0, l1);
                const f16x8 v0h = vfrag(Vh, u, s, 0), v0l = vfrag(Vl, u, s, 0);
                const f16x8 v1h = vfrag(Vh, u, s, 1), v1l = vfrag(Vl, u, s, 1);
                o0 = mfma_f16(v0l, ph, o0);
                o1 = mfma_f16(v1l, ph, o1);
                o0 = mfma_f16(v0h, pl, o0);
                o1 = mfma_f16(v1h, pl, o1);
                o0 = mfma_f16(v0h, ph, o0);
                o1 = mfma_f16(v1h, ph, o1);
            }
    };

    const int nkt = (N + KT - 1) / KT;
    load_kv(0);
    for (int kt = 0; kt + 1 < nkt; ++kt) tile(kt, std::false_type{});
    tile(nkt - 1, std::true_type{});
    __syncthreads();  // the stage is free: reuse it for the O^T transpose

    // Normalise (the 2^10 of p' cancels), transpose O^T through LDS, store whole 256-B head rows.
    const float l_half = l_run[0] + l_run[1];
    const float inv = 1.0f / (l_half + __shfl_xor(l_half, 32));
    float* Os = smem + (wave * 32) * OST;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = o0[4 * g4 + e] * inv; c[e] = o1[4 * g4 + e] * inv; }
        *reinterpret_cast<f32x4*>(&Os[r * OST + 8 * g4 + 4 * h]) = a;
        *reinterpret_cast<f32x4*>(&Os[r * OST + 32 + 8 * g4 + 4 * h]) = c;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int lr = (lane >> 4) + 4 * i, c4 = (lane & 15) * 4;
        const int qrow = q0 + wave * 32 + lr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Os[lr * OST + c4]);
        if (qrow < N) {
            if constexpr (OUT_PLANES) {
                f16x4 hi, lo;
                split4(v * 8.0f, hi, lo);  // K_PLANES_ACT_SCALE
                const int col = head * HD + c4;
                _Float16* o = reinterpret_cast<_Float16*>(out) + (size_t(b) * N + qrow) * 2 * D + (col >> 5) * 64 + (col & 31);
                *reinterpret_cast<f16x4*>(o) = hi;
                *reinterpret_cast<f16x4*>(o + 32) = lo;
            } else {
                *reinterpret_cast<f32x4*>(out + (size_t(b) * N + qrow) * D + head * HD + c4) = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Software-pipelined variant.  Measured on gfx950 (scripts/overlap16_lab.hip): an f16 MFMA phase of one wave
// does NOT overlap a VALU phase of another wave on the same SIMD (2 or 4 waves per SIMD, any priorities or start
// skews: the phases add up), but VALU instructions placed BETWEEN the MFMAs of the same wave cost about 2 cycles
// each instead of 4+ (t ~ 32 + 2 n cycles per MFMA with n VALU ops behind it).  So the only way to hide the
// softmax is inside each wave's own instruction stream: here QK^T of tile t+1 is issued interleaved with the
// max / exp / sum work of tile t, and P.V of tile t interleaved with the hi/lo splitting of P and of the K/V rows
// of tile t+2 (every MFMA is followed by its slice of VALU work and a sched_barrier).  Three K/V stages in LDS:
// tile t+2 is written (under the P.V MFMAs) into the stage nobody reads during iteration t, so an iteration has
// ONE workgroup barrier and no store-only bubble (with two stages: barrier, store, barrier = 17 % of an iteration
// with the matrix pipe idle).  One workgroup of 8 waves per CU (256 registers per wave at 2 waves per SIMD).
#ifdef ATTN_STAMPS  // dev: per-iteration cycle stamps of block 300, wave 0 (scripts/attn_stamps.py)
__device__ unsigned long long g_attn_dbg[32 * 8];
#define ATTN_STAMP(t, slot)                                                                                     \
    do {                                                                                                        \
        if (blockIdx.x == 300 && tid == 0 && (t) < 32) g_attn_dbg[(t) * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define ATTN_STAMP(t, slot) do {} while (0)
#endif
// Diagnostic instantiation only (template flag DIAG; bench.py's `attention_ramp` leg): (wave, 64-key tile) pairs that took the
// exact pass since the last read.  NOT in the product kernel: a single relaxed atomic on the cold path costs the hot loop 16
// spilled registers and 6 % (0.644 -> 0.685 ms, measured) — the loop sits at 256 VGPRs with zero slack.
__device__ unsigned long long g_attn_exact_passes;
constexpr int STAGE_H = 2 * (K_PLANE + V_PLANE);  // halves per K/V stage (Kh | Kl | Vh | Vl)
constexpr size_t X3_ATTN_PIPE_BYTES = size_t(3) * STAGE_H * sizeof(_Float16);  // 129 024 B: three stages, see below
static_assert(X3_ATTN_PIPE_BYTES >= X3_ATTN_EPI_BYTES, "epilogue staging fits the stages");

// IN_PLANES: qkv is the planes tensor the QKV GEMM's epilogue writes (pope_hip.h layout, scale 8): K/V rows go to
// LDS as they are (no per-tile hi/lo split: 48 VALU instructions, 8 LDS stores and a wait on loads issued one
// iteration earlier per wave and tile -> 16-byte copies from registers loaded TWO iterations earlier).
template <bool OUT_PLANES, bool IN_PLANES, bool DIAG = false>
__global__ __launch_bounds__(NT, 2) void attn_f16x3_pipe_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  int N, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_qb = (N + QB - 1) / QB;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = logical / n_qb, head = bh % heads, b = bh / heads, q0 = (logical - bh * n_qb) * QB;
    const int D = heads * HD, rs = 3 * D;
    const float* base = qkv + size_t(b) * N * rs;
    const int koff = D + head * HD;

    // planes carry x8 (K_PLANES_ACT_SCALE): the K factor is folded into Q here, the V factor into the final 1/l
    constexpr float QSCALE = 0.125f * 1.44269504088896340736f * (IN_PLANES ? 0.125f * 0.125f : 1.0f);
    f16x8 qh[4], ql[4];
    {
        const int qrow = q0 + wave * 32 + r;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (qrow < N) {
                if constexpr (IN_PLANES) {  // 8 hi + 8 lo halves of chunk head*2 + kg/2 -> (hi + lo), still x8
                    const _Float16* p = reinterpret_cast<const _Float16*>(base) + size_t(qrow) * 2 * rs +
                                        (head * 2 + (kg >> 1)) * 64 + 16 * (kg & 1) + 8 * h;
                    const f16x8 ph = *reinterpret_cast<const f16x8*>(p), pl = *reinterpret_cast<const f16x8*>(p + 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = float(ph[e]) + float(pl[e]);
                        v1[e] = float(ph[4 + e]) + float(pl[4 + e]);
                    }
                } else {
                    const float* p = base + size_t(qrow) * rs + head * HD + 16 * kg + 8 * h;
                    v0 = *reinterpret_cast<const f32x4*>(p);
                    v1 = *reinterpret_cast<const f32x4*>(p + 4);
                }
            }
            f16x4 h0, l0, h1, l1;
            split4(v0 * QSCALE, h0, l0);
            split4(v1 * QSCALE, h1, l1);
            qh[kg] = cat(h0, h1);
            ql[kg] = cat(l0, l1);
        }
    }

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, unsigned(N) * unsigned(rs) * 4u, 0x00020000);
    constexpr int RPP = NT / 16, NP = KT / RPP;
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    unsigned kvoff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) kvoff[i] = (unsigned(srow + RPP * i) * unsigned(rs) + scol + koff) * 4u;
    const unsigned tile_bytes = unsigned(KT) * unsigned(rs) * 4u, v_delta = unsigned(D) * 4u;
    f32x4 rk[NP], rv[NP];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff[i], kt * tile_bytes, 0));
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff[i] + v_delta, kt * tile_bytes, 0));
        }
    };
    // K/V rows of the tile in flight, already split (done under the P.V MFMAs), and their LDS store
    f16x4 skh[NP], skl[NP], svh[NP], svl[NP];
    auto split_kv = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            split4(rk[i], skh[i], skl[i]);
            split4(rv[i], svh[i], svl[i]);
        }
    };
    auto write_k_row = [&](int st, int i) {
        _Float16* S = lds + st * STAGE_H;
        *reinterpret_cast<f16x4*>(S + (srow + RPP * i) * KST + scol) = skh[i];
        *reinterpret_cast<f16x4*>(S + K_PLANE + (srow + RPP * i) * KST + scol) = skl[i];
    };
    auto write_v_row = [&](int st, int i) {
        _Float16* S = lds + st * STAGE_H;
        *reinterpret_cast<f16x4*>(S + 2 * K_PLANE + (srow + RPP * i) * VST + scol) = svh[i];
        *reinterpret_cast<f16x4*>(S + 2 * K_PLANE + V_PLANE + (srow + RPP * i) * VST + scol) = svl[i];
    };
    auto write_kv = [&](int st) {
        _Float16* S = lds + st * STAGE_H;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            *reinterpret_cast<f16x4*>(S + (srow + RPP * i) * KST + scol) = skh[i];
            *reinterpret_cast<f16x4*>(S + K_PLANE + (srow + RPP * i) * KST + scol) = skl[i];
            *reinterpret_cast<f16x4*>(S + 2 * K_PLANE + (srow + RPP * i) * VST + scol) = svh[i];
            *reinterpret_cast<f16x4*>(S + 2 * K_PLANE + V_PLANE + (srow + RPP * i) * VST + scol) = svl[i];
        }
    };

    // IN_PLANES staging: thread -> (key = tid >> 3, 16-byte piece p = tid & 7 of a 128-byte chunk [32 hi | 32 lo]);
    // four chunks per key (K d 0..31, K d 32..63, V d 0..31, V d 32..63), two register sets = two tiles in flight.
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 pa[4] = {}, pb[4] = {};
    const int pkey = tid >> 3, pp = tid & 7;
    const unsigned pl_off = unsigned(pkey) * unsigned(rs) * 4u + unsigned((D >> 5) + head * 2) * 128u + unsigned(pp) * 16u;
    const unsigned pl_vdelta = unsigned(D >> 5) * 128u;
    auto load_planes = [&](int kt, u32x4 (&st)[4]) {
        // never address a tile past the last one: the descriptor's range check subtracts the scalar tile offset
        // from the extent, which must not go negative (the stale registers are then stored to a stage nobody reads)
        if (kt * KT >= N) return;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            st[c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pl_off + (c >> 1) * pl_vdelta + (c & 1) * 128u, kt * tile_bytes, 0);
    };
    auto write_chunk = [&](int st, int c, const u32x4 (&regs)[4]) {  // c = 0, 1: K; 2, 3: V
        _Float16* S = lds + st * STAGE_H + (c < 2 ? 0 : 2 * K_PLANE) + (pp >> 2) * (c < 2 ? K_PLANE : V_PLANE) +
                      pkey * (c < 2 ? KST : VST) + (c & 1) * 32 + (pp & 3) * 8;
        *reinterpret_cast<u32x4*>(S) = regs[c];
    };

    const int tr_off = (4 * h + ((lane & 15) >> 2)) * VST + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto vfrag = [&](const _Float16* plane, int u, int s, int dt) {
        const _Float16* p = plane + tr_off + (32 * u + 16 * s) * VST + 32 * dt;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
        const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 8 * VST));
        return cat(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, c));
    };

    // output accumulators; two score buffers that swap roles every tile (P = parity of the tile whose probabilities a
    // buffer holds): sb[P] = scores / probabilities of tile t, sb[P ^ 1] = scores of tile t + 1 — no copies
    f32x16 o0, o1, sb[2][2];
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    f32x2 l_run = {0.f, 0.f};
    // the softmax reference of the rows (see below), negated, in every element: the C operand the score MFMAs start from
    f32x16 nref;
#pragma unroll
    for (int i = 0; i < 16; ++i) nref[i] = INFINITY;

    struct KFrag { f16x8 k0h, k0l, k1h, k1l; };
    auto read_kfrag = [&](int st, int kg, KFrag& f) {
        const _Float16* kb_h = lds + st * STAGE_H + r * KST + 8 * h + 16 * kg;
        f.k0h = *reinterpret_cast<const f16x8*>(kb_h);
        f.k0l = *reinterpret_cast<const f16x8*>(kb_h + K_PLANE);
        f.k1h = *reinterpret_cast<const f16x8*>(kb_h + 32 * KST);
        f.k1l = *reinterpret_cast<const f16x8*>(kb_h + K_PLANE + 32 * KST);
    };
    // MFMA j (0..5) of d step kg.  The first MFMA of a chain starts from C = -reference (biased: the scores come out
    // as s - reference, ready for v_exp) or from the inline constant 0 (raw scores): no fill instructions either way
    auto qk_step = [&](int kg, int j, const KFrag& f, f32x16& d0, f32x16& d1, auto biased) {
        const f32x16 zero = {};
        const f32x16 c_init = decltype(biased)::value ? nref : zero;
        if (j == 0) d0 = mfma_f16(f.k0l, qh[kg], kg == 0 ? c_init : d0);
        if (j == 1) d1 = mfma_f16(f.k1l, qh[kg], kg == 0 ? c_init : d1);
        if (j == 2) d0 = mfma_f16(f.k0h, ql[kg], d0);
        if (j == 3) d1 = mfma_f16(f.k1h, ql[kg], d1);
        if (j == 4) d0 = mfma_f16(f.k0h, qh[kg], d0);
        if (j == 5) d1 = mfma_f16(f.k1h, qh[kg], d1);
    };
    auto mask_tail = [&](int kt, f32x16& d0, f32x16& d1) {  // padded keys of the last tile
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = kt * KT + mfma32_row(i, h);
            if (key >= N) d0[i] = -INFINITY;
            if (key + 32 >= N) d1[i] = -INFINITY;
        }
    };
    // Online softmax of (c0, c1) with a LAZY reference (log2 domain): p' = 2^(s - ref), where ref = (the row maximum
    // as of the last advance) - 8, not this tile's.  On gfx950 every VALU instruction issued between the MFMAs takes
    // about two cycles from the matrix pipe (DESIGN.md finding 4), so what can go, goes:
    //   * the unconditional rescale of o and l (21 multiplies) and the cross-half exchange of the row maximum: a tile
    //     whose scores stay below ref + 16 needs neither — p' < 2^16 still splits into f16 (hi, lo), and with 2^8 of
    //     headroom under the running maximum the split keeps fp32 accuracy;
    //   * the subtraction: the score MFMAs start from C = -ref (nref), so s - ref is what they deliver.
    // LOOK-AHEAD (round 4).  Whether a tile fits is decided BEFORE its exponentials, from its biased scores themselves:
    // the scores of tile t + 1 are complete when phase 1 of iteration t ends, and their per-lane maximum (16 v_max3) is
    // taken behind the P.V MFMAs of iteration t.  If no lane reaches 2^16 the tile runs on the old reference; otherwise
    // `advance` — between the iterations, nothing recomputed — takes the true row maximum (one cross-half swap), moves the
    // reference of the rows that need it to (their maximum - 8), rescales their o and l and re-biases the waiting scores.
    // Rounds 2-3 instead detected the overflow AFTER the exponentials (from the tile's probability sum) and then
    // recomputed the tile's scores from its K stage (24 MFMAs) for an exact pass: one more tile's worth of time per
    // event, +49 % on scores that climb by 9 log2 units per tile; now +15 %, and the hot loop has no redo path hanging on
    // its register allocation.  Results do not depend on which path ran beyond fp32 rounding.
    constexpr float LAZY_HEADROOM = 8.0f, LAZY_LIMIT_LOG2 = 15.99f;   // 2^15.99 < 65 504, the largest f16
    f32x2 sm_ls = {0.f, 0.f};
    // true row maximum of the RAW scores (c0, c1) -> new reference; o and l rescaled; returns old - new reference
    auto exact_prepare = [&](const f32x16& c0, const f32x16& c1) __attribute__((always_inline)) {
        float mt = vmax3(c0[0], c1[0], c0[1]);
#pragma unroll
        for (int i = 1; i < 15; ++i) mt = vmax3(mt, c1[i], c0[i + 1]);
        mt = __builtin_fmaxf(mt, c1[15]);   // compiler-generated: feeds the permlane swap
        // the row maximum lives in lanes l and l ^ 32: v_permlane32_swap hands each half the other's value in one
        // instruction (a __shfl_xor is a ds_bpermute: 7 address instructions, an LDS round trip and a wait).  Its
        // operand and its results are touched by COMPILER-generated instructions only: the hazard recognizer does
        // not look into inline asm, and VALU write -> permlane swap -> VALU read need wait states on gfx950.
        float ma, mb;
        pope_xor32_pair(mt, ma, mb);
        const float ref_old = -nref[0];
        const float ref_new = __builtin_fmaxf(ref_old, __builtin_fmaxf(ma, mb) - LAZY_HEADROOM);  // never decreases
        const float delta = ref_old - ref_new;                 // first tile: -inf
        const float alpha = __builtin_amdgcn_exp2f(delta);     // first tile: 0 on o = l = 0
#pragma unroll
        for (int i = 0; i < 16; ++i) nref[i] = -ref_new;
        l_run = l_run * alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
        return delta;
    };
    // two probabilities, in place, adjacent in their tuple (pairing c0[i] with c1[i] makes the register allocator
    // permute the tuples for the v_pk_add and copy them back: 24 v_mov per tile); biased: the scores are s - ref already
    auto exp_pair = [&](int i, f32x16& c0, f32x16& c1, auto biased) __attribute__((always_inline)) {
        f32x16& c = i < 8 ? c0 : c1;
        const int e = 2 * (i & 7);
        if constexpr (decltype(biased)::value) {
            c[e] = __builtin_amdgcn_exp2f(c[e]);
            c[e + 1] = __builtin_amdgcn_exp2f(c[e + 1]);
        } else {
            c[e] = __builtin_amdgcn_exp2f(c[e] + nref[0]);
            c[e + 1] = __builtin_amdgcn_exp2f(c[e + 1] + nref[0]);
        }
        // two plain adds, not one v_pk_add_f32: a packed fp32 instruction beside MFMAs costs 22 - 26 cycles more than its issue
        // slot (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); the same two sums, bit for bit
        float a0 = sm_ls[0], a1 = sm_ls[1];
        a0 += c[e];
        // pin the running sums to their slot: the optimiser otherwise sinks the whole (dependent) chain of adds out of
        // the MFMA shadow to the top of the next iteration, right behind the barrier, one s_nop per add (and re-packs them)
        asm volatile("" : "+v"(a0));
        a1 += c[e + 1];
        asm volatile("" : "+v"(a1));
        sm_ls = f32x2{a0, a1};
    };
    // slice `slot` (0..23) of the fast pass: two probabilities behind two of every three MFMAs.  (Measured and dropped:
    // four pairs in front of the first MFMA, under the latency of its K fragments; 2, 3 or 6 pairs behind every 4th,
    // 6th or 12th MFMA; both halves of a P split behind one MFMA — all 0.5 to 2 % slower than the even spread.)
    auto softmax_slice = [&](int slot, auto ptag) __attribute__((always_inline)) {
        if (slot == 0) sm_ls = f32x2{0.f, 0.f};
        if (slot % 3 != 2)
            exp_pair((slot / 3) * 2 + slot % 3, sb[decltype(ptag)::value][0], sb[decltype(ptag)::value][1], std::true_type{});
    };
    // per-lane maximum of the waiting (biased) scores, one v_max3 per slot (two values each); slot 0 starts the chain
    float la_mt = 0.f;
    auto lookahead_slice = [&](int slot, const f32x16& n0, const f32x16& n1) __attribute__((always_inline)) {
        if (slot == 0) la_mt = vmax3(n0[0], n1[0], n0[1]);
        else if (slot < 15) la_mt = vmax3(la_mt, n1[slot], n0[slot + 1]);
        else if (slot == 15) la_mt = __builtin_fmaxf(la_mt, n1[15]);   // compiler-generated: feeds the ballot / the swap
    };
    // the waiting scores (n0, n1) reach 2^16 under the current reference somewhere in this wave: move the reference of
    // those rows.  Cold path; everything is changed IN PLACE (no value is handed back through new registers).
    auto advance = [&](f32x16& n0, f32x16& n1) __attribute__((always_inline)) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(la_mt < LAZY_LIMIT_LOG2)) != 0, 0)) {
            if constexpr (DIAG)
                if (lane == 0) (void)__hip_atomic_fetch_add(&g_attn_exact_passes, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float ma, mb;
            pope_xor32_pair(la_mt, ma, mb);                       // the row lives in lanes l and l ^ 32
            const float over = __builtin_fmaxf(__builtin_fmaxf(ma, mb) - LAZY_HEADROOM, 0.f);   // rows that fit: 0
            const float delta = -over, alpha = __builtin_amdgcn_exp2f(delta);                  // alpha = 1 for them
            const float nr = nref[0] + delta;                     // -(ref + over)
#pragma unroll
            for (int i = 0; i < 16; ++i) nref[i] = nr;
            l_run = l_run * alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
#pragma unroll
            for (int i = 0; i < 16; ++i) { n0[i] += delta; n1[i] += delta; }
        }
    };
    // after the softmax slices of a tile: its probability sum joins the running denominator
    auto settle = [&]() __attribute__((always_inline)) { l_run += sm_ls; };
    // ---- phase 1: S^T(tile in stage st_next) -> (n0, n1), each MFMA followed by a slice of the softmax of (c0, c1)
    auto phase1 = [&](int st_next, auto ptag) {
        f32x16& n0 = sb[decltype(ptag)::value ^ 1][0];
        f32x16& n1 = sb[decltype(ptag)::value ^ 1][1];
        KFrag kf[2];
        read_kfrag(st_next, 0, kf[0]);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int kg = i / 6, j = i % 6;
            if (j == 0 && kg < 3) read_kfrag(st_next, kg + 1, kf[(kg + 1) & 1]);
            qk_step(kg, j, kf[kg & 1], n0, n1, std::true_type{});
            softmax_slice(i, ptag);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto softmax_only = [&](auto ptag) {
#pragma unroll
        for (int i = 0; i < 24; ++i) softmax_slice(i, ptag);
    };
    // ---- phase 2: O^T += V^T(stage st) . P^T with P = (c0, c1); the hi/lo split of the next 16 keys' probabilities
    // (and, SPLIT_KV, of one K/V row of tile t+2 per group) sits behind the MFMAs of the current 16 keys
    auto split_group = [&](int g, int half, f16x4& hi, f16x4& lo, auto ptag) {
        const int u = g >> 1, s2 = g & 1;
        f32x4 pv4;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv4[e] = sb[decltype(ptag)::value][u][8 * s2 + 4 * half + e];
        split4(pv4, hi, lo);
    };
    auto phase2 = [&](int st, int st_write, auto split_tag, const u32x4 (&pregs)[4], auto ptag, auto look_tag) {
        constexpr bool SPLIT_KV = decltype(split_tag)::value;
        constexpr bool LOOK = decltype(look_tag)::value;   // the scores of the next tile wait in the other buffer
        const _Float16* Vh = lds + st * STAGE_H + 2 * K_PLANE;
        const _Float16* Vl = Vh + V_PLANE;
        f16x4 h0[2], l0[2], h1[2], l1[2];
        // V fragments: the hi planes double-buffered one 16-key group ahead; the lo planes (needed by the first two
        // MFMAs of a group only) in ONE buffer that is refilled for the next group right behind those two — 8 VGPRs
        // less than double-buffering both, which is what keeps the tile loads in flight out of scratch
        f16x8 vh[2][2], vl[2];
        split_group(0, 0, h0[0], l0[0], ptag);
        split_group(0, 1, h1[0], l1[0], ptag);
        vh[0][0] = vfrag(Vh, 0, 0, 0); vl[0] = vfrag(Vl, 0, 0, 0);
        vh[0][1] = vfrag(Vh, 0, 0, 1); vl[1] = vfrag(Vl, 0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int g = i / 6, j = i % 6, cur = g & 1, nxt = cur ^ 1;
            const int u = (g + 1) >> 1, s2 = (g + 1) & 1;
            if (j == 0 && g < 3) { vh[nxt][0] = vfrag(Vh, u, s2, 0); vh[nxt][1] = vfrag(Vh, u, s2, 1); }
            const f16x8 ph = cat(h0[cur], h1[cur]), pl = cat(l0[cur], l1[cur]);
            if (j == 0) o0 = mfma_f16(vl[0], ph, o0);
            if (j == 1) o1 = mfma_f16(vl[1], ph, o1);
            if (j == 2) o0 = mfma_f16(vh[cur][0], pl, o0);
            if (j == 3) o1 = mfma_f16(vh[cur][1], pl, o1);
            if (j == 4) o0 = mfma_f16(vh[cur][0], ph, o0);
            if (j == 5) o1 = mfma_f16(vh[cur][1], ph, o1);
            if (j == 2 && g < 3) { vl[0] = vfrag(Vl, u, s2, 0); vl[1] = vfrag(Vl, u, s2, 1); }
            if (g < 3) {
                if (j == 1) split_group(g + 1, 0, h0[nxt], l0[nxt], ptag);
                if (j == 3) split_group(g + 1, 1, h1[nxt], l1[nxt], ptag);
            }
            if (LOOK && i % 3 != 1) lookahead_slice(i - (i + 1) / 3, sb[decltype(ptag)::value ^ 1][0], sb[decltype(ptag)::value ^ 1][1]);
            if (SPLIT_KV && j == 5) {  // a quarter of tile t+2's staging per 16-key group
                if constexpr (IN_PLANES) {
                    write_chunk(st_write, g, pregs);
                } else {
                    const int i2 = g >> 1;
                    if (g & 1) { split4(rv[i2], svh[i2], svl[i2]); write_v_row(st_write, i2); }
                    else { split4(rk[i2], skh[i2], skl[i2]); write_k_row(st_write, i2); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nkt = (N + KT - 1) / KT;
    // prologue: tile 0 -> stage 0, S^T(0); tile 1 -> stage 1; loads of tile 2 in flight
    if constexpr (IN_PLANES) {
        load_planes(0, pa);
        load_planes(1, pb);
#pragma unroll
        for (int c = 0; c < 4; ++c) write_chunk(0, c, pa);
        load_planes(2, pa);
    } else {
        load_kv(0);
        split_kv();
        write_kv(0);
        if (nkt > 1) load_kv(1);
    }
    __syncthreads();
    {
        KFrag kf;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            read_kfrag(0, kg, kf);
#pragma unroll
            for (int j = 0; j < 6; ++j) qk_step(kg, j, kf, sb[0][0], sb[0][1], std::false_type{});
        }
    }
    if (nkt == 1) mask_tail(0, sb[0][0], sb[0][1]);
    if constexpr (IN_PLANES) {
#pragma unroll
        for (int c = 0; c < 4; ++c) write_chunk(1, c, pb);
        load_planes(3, pb);  // now: pa = tile 2, pb = tile 3
    } else if (nkt > 1) {
        split_kv();
        write_kv(1);
        if (nkt > 2) load_kv(2);
    }
    __syncthreads();
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sb[0][0]), "+v"(sb[0][1]));  // XDL write -> asm VALU read (vmax3) wait states
    exact_prepare(sb[0][0], sb[0][1]);  // tile 0 sets the first reference; its raw scores are shifted here, once
#pragma unroll
    for (int i = 0; i < 16; ++i) { sb[0][0][i] += nref[0]; sb[0][1][i] += nref[0]; }

    // stage of tile t = t % 3
    int st_cur = 0, st_next = 1, st_write = 2, t = 0;
    auto rotate = [&]() { const int x = st_cur; st_cur = st_next; st_next = st_write; st_write = x; };
    // one steady-state iteration (tiles t+1 and t+2 exist); `regs` holds tile t+2 and is refilled with tile t+4;
    // ptag = parity of t = the score buffer that holds tile t
    auto steady = [&](u32x4 (&regs)[4], auto ptag) {
        ATTN_STAMP(t, 0);
        phase1(st_next, ptag);
        settle();
        ATTN_STAMP(t, 1);
        // also moves tile t+2 from registers into stage st_write, and takes the look-ahead maximum of tile t+1's scores
        phase2(st_cur, st_write, std::true_type{}, regs, ptag, std::true_type{});
        advance(sb[decltype(ptag)::value ^ 1][0], sb[decltype(ptag)::value ^ 1][1]);
        if constexpr (IN_PLANES) load_planes(t + 4, regs);
        else if (t + 3 < nkt) load_kv(t + 3);
        ATTN_STAMP(t, 2);
        __syncthreads();  // tile t+2 is published; every wave is done with stage st_cur
        ATTN_STAMP(t, 3);
        rotate();
        ++t;
    };
    while (t + 2 < nkt) {
        steady(pa, P0{});
        if (t + 2 < nkt) steady(pb, P1{});
    }
    // the last one or two tiles; ptag = parity of t
    auto tail = [&](auto ptag) {
        constexpr int P = decltype(ptag)::value;
        using Q = std::integral_constant<int, P ^ 1>;
        if (t + 1 < nkt) {  // second-to-last tile: S^T of the last tile needs the key mask
            phase1(st_next, ptag);
            mask_tail(t + 1, sb[P ^ 1][0], sb[P ^ 1][1]);
            asm volatile("" : "+v"(sb[P ^ 1][0]), "+v"(sb[P ^ 1][1]));
            settle();
            phase2(st_cur, st_write, std::false_type{}, pa, ptag, std::true_type{});   // look-ahead over the masked scores
            advance(sb[P ^ 1][0], sb[P ^ 1][1]);
            rotate();
            ++t;
            softmax_only(Q{});  // last tile
            settle();
            phase2(st_cur, st_write, std::false_type{}, pa, Q{}, std::false_type{});
        } else {
            softmax_only(ptag);
            settle();
            phase2(st_cur, st_write, std::false_type{}, pa, ptag, std::false_type{});
        }
    };
    if (t & 1) tail(P1{});
    else tail(P0{});
    __syncthreads();  // the stages are free: reuse them for the O^T transpose

    // Normalise (the 2^8 of p' cancels), transpose O^T through LDS, store whole 256-B head rows.  The lane coordinates
    // are derived again from threadIdx behind an asm fence: kept live across the main loop they are what gets spilled
    // (the loop runs at 256 VGPRs) — a scratch round trip per thread for four registers.
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6, r_e = lane_e & 31, h_e = lane_e >> 5;
    const float l_half = l_run[0] + l_run[1];
    const float inv = (IN_PLANES ? 0.125f : 1.0f) / (l_half + __shfl_xor(l_half, 32));  // V planes carry x8
    float* Os = smem + (wave_e * 32) * OST;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = o0[4 * g4 + e] * inv; c[e] = o1[4 * g4 + e] * inv; }
        *reinterpret_cast<f32x4*>(&Os[r_e * OST + 8 * g4 + 4 * h_e]) = a;
        *reinterpret_cast<f32x4*>(&Os[r_e * OST + 32 + 8 * g4 + 4 * h_e]) = c;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int lr = (lane_e >> 4) + 4 * i, c4 = (lane_e & 15) * 4;
        const int qrow = q0 + wave_e * 32 + lr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Os[lr * OST + c4]);
        if (qrow < N) {
            if constexpr (OUT_PLANES) {
                f16x4 hi, lo;
                split4(v * 8.0f, hi, lo);  // K_PLANES_ACT_SCALE
                const int col = head * HD + c4;
                _Float16* o = reinterpret_cast<_Float16*>(out) + (size_t(b) * N + qrow) * 2 * D + (col >> 5) * 64 + (col & 31);
                // written once, read once (by the proj GEMM, whose A loads carry the same hint): proj -2 %
                __builtin_nontemporal_store(hi, reinterpret_cast<f16x4*>(o));
                __builtin_nontemporal_store(lo, reinterpret_cast<f16x4*>(o + 32));
            } else {
                *reinterpret_cast<f32x4*>(out + (size_t(b) * N + qrow) * D + head * HD + c4) = v;
            }
        }
    }
}

}  // namespace

static_assert(K_PLANES_ACT_SCALE == 8.0f, "attention planes epilogue scale");

template <bool OUT_PLANES>
static int launch_attn_x3(const float* qkv, float* out, int B, int N, int heads, hipStream_t stream) {
    if (B <= 0 || N <= 0 || heads <= 0 || size_t(B) * heads * ((N + QB - 1) / QB) > 0x7fffffffull) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return POPE_ERR_ARG;
    if (size_t(N) * 3 * heads * HD * 4 >= (size_t(1) << 32)) return POPE_ERR_ARG;
    const dim3 grid(unsigned((N + QB - 1) / QB) * heads * B);
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(attn_f16x3_pipe_kernel<OUT_PLANES, false>, X3_ATTN_PIPE_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_f16x3_pipe_kernel<OUT_PLANES, false>), grid, dim3(NT), X3_ATTN_PIPE_BYTES, stream, qkv, out, N, heads);
    return pope_check_launch();
}

#ifdef ATTN_STAMPS
extern "C" int pope_lab_attn_stamps(unsigned long long* host256) {
    return hipMemcpyFromSymbol(host256, HIP_SYMBOL(g_attn_dbg), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1;
}
#endif

int pope_launch_attention_f16x3(const float* qkv, float* out, int B, int N, int heads, hipStream_t stream) {
    return launch_attn_x3<false>(qkv, out, B, N, heads, stream);
}
// the diagnostic twin of pope_launch_attention_f16x3_planes_io: same results, counts exact passes (slower: see above)
int pope_launch_attention_f16x3_planes_io_diag(const void* qkv_planes, void* out_planes, int B, int N, int heads, long long* exact_passes_host,
                                               hipStream_t stream) {
    if (B <= 0 || N <= 0 || heads <= 0 || ((heads * HD) & 31) || size_t(B) * heads * ((N + QB - 1) / QB) > 0x7fffffffull) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(qkv_planes) & 15) || (reinterpret_cast<uintptr_t>(out_planes) & 15) || !exact_passes_host) return POPE_ERR_ARG;
    if (size_t(N) * 3 * heads * HD * 4 >= (size_t(1) << 32)) return POPE_ERR_ARG;
    const dim3 grid(unsigned((N + QB - 1) / QB) * heads * B);
    static pope_dev_mask lds_ok{0};
    if (!pope_opt_in_lds(attn_f16x3_pipe_kernel<true, true, true>, X3_ATTN_PIPE_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    unsigned long long v = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_exact_passes), &v, sizeof(v)) != hipSuccess) return POPE_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_f16x3_pipe_kernel<true, true, true>), grid, dim3(NT), X3_ATTN_PIPE_BYTES, stream,
                       static_cast<const float*>(qkv_planes), static_cast<float*>(out_planes), N, heads);
    if (pope_check_launch() || hipStreamSynchronize(stream) != hipSuccess) return POPE_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_attn_exact_passes), sizeof(v)) != hipSuccess) return POPE_ERR_LAUNCH;
    *exact_passes_host = (long long)v;
    return POPE_OK;
}
int pope_launch_attention_f16x3_planes_io(const void* qkv_planes, void* out_planes, int B, int N, int heads, hipStream_t stream) {
    if (B <= 0 || N <= 0 || heads <= 0 || ((heads * HD) & 31) || size_t(B) * heads * ((N + QB - 1) / QB) > 0x7fffffffull) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(qkv_planes) & 15) || (reinterpret_cast<uintptr_t>(out_planes) & 15)) return POPE_ERR_ARG;
    if (size_t(N) * 3 * heads * HD * 4 >= (size_t(1) << 32)) return POPE_ERR_ARG;
    const dim3 grid(unsigned((N + QB - 1) / QB) * heads * B);
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(attn_f16x3_pipe_kernel<true, true>, X3_ATTN_PIPE_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    hipLaunchKernelGGL((attn_f16x3_pipe_kernel<true, true>), grid, dim3(NT), X3_ATTN_PIPE_BYTES, stream,
                       static_cast<const float*>(qkv_planes), static_cast<float*>(out_planes), N, heads);
    return pope_check_launch();
}


int pope_launch_attention_f16x3_planes(const float* qkv, void* out_planes, int B, int N, int heads, hipStream_t stream) {
    if ((heads * HD) & 31) return POPE_ERR_ARG;
    return launch_attn_x3<true>(qkv, static_cast<float*>(out_planes), B, N, heads, stream);
}
