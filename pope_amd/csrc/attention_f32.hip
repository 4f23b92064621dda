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

namespace {

constexpr int HD = 64;    // head dim (all DINOv2 archs)
constexpr int KT = 64;    // keys per LDS tile
constexpr int QB = 128;   // queries per block (4 waves x 32)
constexpr int ST = 68;    // padded LDS row (floats): 17 x 16 B -> ds_read_b128 conflict-free
// ONE K and ONE V stage (34.8 KB): three workgroups per CU, so every SIMD hosts three waves of
// independent workgroups whose softmax (VALU) and barrier phases fall under each other's MFMAs.
constexpr size_t ATTN_LDS_BYTES = size_t(2) * KT * ST * sizeof(float);

// ABLATE is 0 in the product; the lab harness (scripts/attn_lab.hip) instantiates timing-only
// variants: bit0 = no softmax VALU, bit1 = no P.V MFMAs, bit2 = no Q.K^T MFMAs.
template <int ABLATE>
__global__ __launch_bounds__(256, 3) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int N, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;            // [KT][ST]
    float* Vs = smem + KT * ST;  // [KT][ST]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * QB;
    const int D = heads * HD, rs = 3 * D;
    const float* base = qkv + size_t(b) * N * rs;
    const int koff = D + head * HD, voff = 2 * D + head * HD;

    // Q^T fragments for all 32 k-steps, pre-scaled by head_dim^-0.5 = 0.125 (exact).
    f32x4 q[8];
    {
        const int qrow = q0 + wave * 32 + r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (qrow < N) v = *reinterpret_cast<const f32x4*>(base + size_t(qrow) * rs + head * HD + 8 * j + 4 * h);
            q[j] = v * 0.125f;
        }
    }

    const int srow = tid >> 4, scol = (tid & 15) * 4;
    f32x4 rk[4], rv[4];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = kt * KT + srow + 16 * i;
            f32x4 zk = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f};
            if (key < N) {
                const float* p = base + size_t(key) * rs + scol;
                zk = *reinterpret_cast<const f32x4*>(p + koff);
                zv = *reinterpret_cast<const f32x4*>(p + voff);
            }
            rk[i] = zk;
            rv[i] = zv;
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
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (N + KT - 1) / KT;
    load_kv(0);
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt) __syncthreads();  // every wave is done with the previous K/V stage
        store_kv();
        __syncthreads();
        if (kt + 1 < nkt) load_kv(kt + 1);  // in flight under this tile's 128 MFMAs

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
        if (kt == nkt - 1) {  // mask the padded keys of the last tile (wave-uniform branch)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kt * KT + mfma32_row(i, h);
                if (key >= N) s0[i] = -INFINITY;
                if (key + 32 >= N) s1[i] = -INFINITY;
            }
        }
        if constexpr (!(ABLATE & 1)) {
        float mt = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mt = fmaxf(mt, fmaxf(s0[i], s1[i]));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float ls = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = __expf(s0[i] - m_new);
            s1[i] = __expf(s1[i] - m_new);
            ls += s0[i] + s1[i];
        }
        l_run = l_run * alpha + ls;
        o0 *= alpha;
        o1 *= alpha;
        } else {
            l_run = 1.f;
        }

        const float* vb = &Vs[r];
        if constexpr (ABLATE & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] += s0[i]; o1[i] += s1[i]; }
        } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = mfma32_row(t, h);
            o0 = mfma_32x32x2(vb[row * ST], s0[t], o0);
            o1 = mfma_32x32x2(vb[row * ST + 32], s0[t], o1);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = 32 + mfma32_row(t, h);
            o0 = mfma_32x32x2(vb[row * ST], s1[t], o0);
            o1 = mfma_32x32x2(vb[row * ST + 32], s1[t], o1);
        }
        }
    }
    __syncthreads();  // the stage is free: reuse it for the O^T transpose

    // Normalise, transpose O^T through LDS (K buffers are free after the final barrier; each
    // wave touches only its own 32 rows) and store whole 256-B head rows.
    const float l_tot = l_run + __shfl_xor(l_run, 32);
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
    if (B <= 0 || N <= 0 || heads <= 0 || B > 65535 || heads > 65535) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return POPE_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_kernel<0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(ATTN_LDS_BYTES)) != hipSuccess)
            return POPE_ERR_LAUNCH;
        attr_set = true;
    }
    const dim3 grid((N + QB - 1) / QB, heads, B);
    hipLaunchKernelGGL(attn_f32_kernel<0>, grid, dim3(256), ATTN_LDS_BYTES, stream, qkv, out, N, heads);
    return pope_check_launch();
}
