// Flash-style multi-head attention in exact fp32 on the gfx950 f32 MFMA.
//
// Reference semantics (dinov2/dinov2/layers/attention.py:49-62): qkv[B,N,3,H,64];
// q*0.125; softmax(q k^T, -1) v; heads re-interleaved to [B,N,H*64].  The N x N score matrix
// (3.6 GB for 64 images at N=1531) is never materialised: per 128-query block the kernel
// streams 64-key K/V tiles through LDS with an online softmax.
//
// MFMA orientation (v_mfma_f32_32x32x2_f32, C/D: column = lane&31, rows in registers):
//   S^T = K . Q^T   -> lane (r,h) holds 16 keys of query r: reductions over keys are
//                      register-local + one cross-half shuffle (no LDS round trip);
//   O^T += V^T . P^T -> the S^T accumulator registers ARE the B operand (key index = the
//                      k of the MFMA, one key per lane half per step); V rows are read from
//                      LDS with the matching key order; alpha/normaliser are per-lane scalars.
#include "common.h"
#include "kernels.h"
#include <type_traits>

namespace {

constexpr int HD = 64;    // head dim (all DINOv2 archs)
constexpr int KT = 64;    // keys per LDS tile
constexpr int QB = 128;   // queries per block (4 waves x 32)
constexpr int ST = 68;    // padded LDS row (floats): 17 x 16 B -> ds_read_b128 conflict-free
// ONE K and ONE V stage (34.8 KB): three workgroups per CU, so every SIMD hosts three waves of
// independent workgroups whose softmax (VALU) and barrier phases fall under each other's MFMAs.
constexpr size_t ATTN_LDS_BYTES = size_t(2) * KT * ST * sizeof(float);

// Single-instruction VALU helpers: hipcc would otherwise canonicalise MFMA outputs before fmaxf
// (one extra v_max per element) and leave most of the packed-f32 forms unused.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// ABLATE is 0 in the product; the lab harness (scripts/attn_lab.hip) instantiates timing-only
// variants: bit0 = no softmax VALU, bit1 = no P.V MFMAs, bit2 = no Q.K^T MFMAs.
template <int ABLATE>
__global__ __launch_bounds__(256, 2) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int N, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;            // [KT][ST]
    float* Vs = smem + KT * ST;  // [KT][ST]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware: the query blocks of one (image, head) get consecutive logical ids, i.e.
    // run on ONE XCD, so its private L2 serves their K/V re-reads (12 query blocks re-read the
    // same 784 KB; spread round-robin over the 8 XCDs the fabric saw 4.6x the algorithmic bytes).
    const int n_qb = (N + QB - 1) / QB;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = logical / n_qb, head = bh % heads, b = bh / heads, q0 = (logical - bh * n_qb) * QB;
    const int D = heads * HD, rs = 3 * D;
    const float* base = qkv + size_t(b) * N * rs;
    const int koff = D + head * HD;

    // Q^T fragments for all 32 k-steps, pre-scaled by head_dim^-0.5 * log2(e): the scores come
    // out of the MFMA already in the log2 domain, so p = exp2(s - m) is one v_sub + one v_exp_f32.
    // (On gfx950 the f32 MFMA executes on the SIMD's f32 VALU lanes — measured: every VALU
    // instruction issued on a SIMD costs its full issue time against the MFMA stream, whichever
    // wave issues it — so the softmax is written for minimum VALU instruction count: packed
    // sub/add/mul, max3, no copies.)
    constexpr float QSCALE = 0.125f * 1.44269504088896340736f;
    f32x4 q[8];
    {
        const int qrow = q0 + wave * 32 + r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (qrow < N) v = *reinterpret_cast<const f32x4*>(base + size_t(qrow) * rs + head * HD + 8 * j + 4 * h);
            q[j] = v * QSCALE;
        }
    }

    // K/V staging: bounds-checked buffer loads (keys >= N read as zeros): one instruction per
    // 16-byte piece, no address arithmetic and no branches in the loop.
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, unsigned(N) * unsigned(rs) * 4u, 0x00020000);
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    unsigned kvoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kvoff[i] = (unsigned(srow + 16 * i) * unsigned(rs) + scol + koff) * 4u;
    const unsigned tile_bytes = unsigned(KT) * unsigned(rs) * 4u, v_delta = unsigned(D) * 4u;
    f32x4 rk[4], rv[4];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff[i], kt * tile_bytes, 0));
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff[i] + v_delta, kt * tile_bytes, 0));
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&Ks[(srow + 16 * i) * ST + scol]) = rk[i];
            *reinterpret_cast<f32x4*>(&Vs[(srow + 16 * i) * ST + scol]) = rv[i];
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY;   // running max, log2 domain
    f32x2 l_run = {0.f, 0.f};  // running sum, two partial lanes (packed adds)

    // V-row base pointers (key rows 8g + 4h of each 32-key half): loop-invariant; laundered through
    // an empty asm so the compiler keeps them in registers instead of re-deriving them with ~18
    // v_add_u32 per tile (every VALU instruction is paid against the MFMA stream).
    typedef __attribute__((address_space(3))) const float* lds_cptr;  // keep the pointers LDS-typed (ds_read, not flat)
    lds_cptr vbase[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        vbase[g] = (lds_cptr)(&Vs[((g >> 2) * 32 + 8 * (g & 3) + 4 * h) * ST + r]);
        asm volatile("" : "+v"(vbase[g]));
    }

    auto tile = [&](int kt, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        if (kt) __syncthreads();  // every wave is done with the previous K/V stage
        store_kv();
        __syncthreads();
        if constexpr (!LAST) load_kv(kt + 1);  // in flight under this tile's 128 MFMAs

        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        const float* kb = &Ks[r * ST + 4 * h];
#pragma unroll
        for (int j = 0; j < ((ABLATE & 4) ? 1 : 8); ++j) {
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(kb + 8 * j);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(kb + 32 * ST + 8 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s0 = mfma_32x32x2(k0[s], q[j][s], s0);
                s1 = mfma_32x32x2(k1[s], q[j][s], s1);
            }
        }
        if constexpr (LAST) {  // mask the padded keys of the last tile
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kt * KT + mfma32_row(i, h);
                if (key >= N) s0[i] = -INFINITY;
                if (key + 32 >= N) s1[i] = -INFINITY;
            }
        }
        if constexpr (!(ABLATE & 1)) {
            // the MFMA results feed inline-asm VALU ops next: cover the XDL-write -> VALU-read wait
            // states ourselves (hipcc pads nothing it cannot see inside an asm statement)
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s0), "+v"(s1));
            float mt = vmax3(s0[0], s1[0], s0[1]);
#pragma unroll
            for (int i = 1; i < 15; ++i) mt = vmax3(mt, s1[i], s0[i + 1]);
            mt = vmax3(mt, s1[15], __shfl_xor(vmax3(mt, s1[15], s1[15]), 32));
            const float m_new = __builtin_fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            const f32x2 mm = {m_new, m_new};
            f32x2 ls = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 a = pk_sub(f32x2{s0[i], s0[i + 1]}, mm), b = pk_sub(f32x2{s1[i], s1[i + 1]}, mm);
                a[0] = __builtin_amdgcn_exp2f(a[0]);
                a[1] = __builtin_amdgcn_exp2f(a[1]);
                b[0] = __builtin_amdgcn_exp2f(b[0]);
                b[1] = __builtin_amdgcn_exp2f(b[1]);
                s0[i] = a[0]; s0[i + 1] = a[1];
                s1[i] = b[0]; s1[i + 1] = b[1];
                ls = pk_add(ls, pk_add(a, b));
            }
            l_run = l_run * alpha + ls;
            o0 *= alpha;  // v_pk_mul_f32 x8
            o1 *= alpha;
        } else {
            l_run = f32x2{0.5f, 0.5f};
        }

        if constexpr (ABLATE & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] += s0[i]; o1[i] += s1[i]; }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                lds_cptr vb = vbase[t >> 2] + (t & 3) * ST;
                o0 = mfma_32x32x2(vb[0], s0[t], o0);
                o1 = mfma_32x32x2(vb[32], s0[t], o1);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                lds_cptr vb = vbase[4 + (t >> 2)] + (t & 3) * ST;
                o0 = mfma_32x32x2(vb[0], s1[t], o0);
                o1 = mfma_32x32x2(vb[32], s1[t], o1);
            }
        }
    };

    const int nkt = (N + KT - 1) / KT;
    load_kv(0);
    for (int kt = 0; kt + 1 < nkt; ++kt) tile(kt, std::false_type{});
    tile(nkt - 1, std::true_type{});
    __syncthreads();  // the stage is free: reuse it for the O^T transpose

    // Normalise, transpose O^T through LDS (K buffers are free after the final barrier; each
    // wave touches only its own 32 rows) and store whole 256-B head rows.
    const float l_half = l_run[0] + l_run[1];
    const float l_tot = l_half + __shfl_xor(l_half, 32);
    const float inv = 1.0f / l_tot;
    float* Os = smem + (wave * 32) * ST;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = o0[4 * g4 + e] * inv; c[e] = o1[4 * g4 + e] * inv; }
        *reinterpret_cast<f32x4*>(&Os[r * ST + 8 * g4 + 4 * h]) = a;
        *reinterpret_cast<f32x4*>(&Os[r * ST + 32 + 8 * g4 + 4 * h]) = c;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int lr = (lane >> 4) + 4 * i, c4 = (lane & 15) * 4;
        const int qrow = q0 + wave * 32 + lr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Os[lr * ST + c4]);
        if (qrow < N) *reinterpret_cast<f32x4*>(out + (size_t(b) * N + qrow) * D + head * HD + c4) = v;
    }
}

}  // namespace

int pope_launch_attention_f32(const float* qkv, float* out, int B, int N, int heads, hipStream_t stream) {
    if (B <= 0 || N <= 0 || heads <= 0 || size_t(B) * heads * ((N + QB - 1) / QB) > 0x7fffffffull) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return POPE_ERR_ARG;
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(attn_f32_kernel<0>, ATTN_LDS_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    const dim3 grid(unsigned((N + QB - 1) / QB) * heads * B);
    hipLaunchKernelGGL(attn_f32_kernel<0>, grid, dim3(256), ATTN_LDS_BYTES, stream, qkv, out, N, heads);
    return pope_check_launch();
}
