// LAB (not product): attention_f16.hip with 64 queries per wave (NQ = 2), four waves = one per SIMD (up to 512 registers), every
// K / V fragment feeding two MFMAs (half the LDS reads per MFMA).  Everything the compiler must see it sees: K / V staged through
// registers (buffer_load -> ds_write), fragment reads through ordinary loads and the ds_read_tr16_b64 builtin — no inline-asm
// reads (the first attempt, with asm reads at 512 registers, computed on registers the allocator had moved), no LDS-direct loads
// (the builtin transposing read waits for vmcnt(0) when any is in flight).  Built into a lab library by scripts/attn_f16_q64_lab.sh.
#include "../pope_amd/csrc/common.h"
#include "../pope_amd/csrc/kernels.h"
#include <type_traits>

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int HD = 64, KT = 64;
constexpr int WAVES = 4, NQ = 2;
constexpr int QB = 32 * NQ * WAVES, NT = 64 * WAVES;    // 256 queries per workgroup
constexpr int KST = 72, VST = 96;                       // halves per LDS row: 144 B (9 pieces), 192 B (12 pieces)
constexpr int K_BYTES = KT * KST * 2, V_BYTES = KT * VST * 2, STAGE_BYTES = K_BYTES + V_BYTES;   // 9 216 + 12 288
constexpr int NST = 4;
constexpr int OST = 68;                                 // epilogue staging row (floats)
constexpr size_t F16_ATTN_LDS = size_t(NST) * STAGE_BYTES;   // 86 016 B
static_assert(size_t(32) * WAVES * OST * sizeof(float) <= F16_ATTN_LDS, "epilogue staging fits the stages");
static_assert(K_BYTES % 1024 == 0 && V_BYTES % 1024 == 0, "whole 1 KB staging instructions per plane");
constexpr int KBLK = K_BYTES / 1024, VBLK = V_BYTES / 1024;   // 9 + 12 = 21 wave-instructions per tile
constexpr int NDMA = (KBLK + VBLK + WAVES - 1) / WAVES;       // six per wave (the three slots past the end repeat blocks 0..2)

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16x8 cat(f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// workgroup barrier that leaves LDS-direct loads in flight (the waits are explicit at the call sites); the empty asm statements
// keep the compiler from moving LDS accesses across it
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(NT) void attn_f16_dma_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, int N, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n_qb = (N + QB - 1) / QB;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);   // query blocks of one (image, head) share an XCD's L2
    const int bh = logical / n_qb, head = bh % heads, b = bh / heads, q0 = (logical - bh * n_qb) * QB;
    const int D = heads * HD, rs = 3 * D;                    // row of the qkv tensor, in halves
    const _Float16* base = qkv + size_t(b) * N * rs;

    // Q^T fragments (B operand of S^T = K . Q^T): lane (r, h) holds Q[q = r][d = 16 kg + 8 h + j] of each of its query blocks
    f16x8 qf[NQ][4];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
        const int qrow = q0 + (wave * NQ + qb) * 32 + r;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            qf[qb][kg] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (qrow < N) qf[qb][kg] = *reinterpret_cast<const f16x8*>(base + size_t(qrow) * rs + head * HD + 16 * kg + 8 * h);
        }
    }

    // ---- staging: block c (0..20) of a tile = 1 KB of the K plane (c < 9) or of the V plane; piece p = 64 jb + lane of the
    // plane lies in row p / ppr at position p % ppr (ppr = 9, 12); positions < 8 are the row's 128-byte line
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, unsigned(N) * unsigned(rs) * 2u, 0x00020000);
    const unsigned tile_bytes = unsigned(KT) * unsigned(rs) * 2u;
    // register staging: 1 024 pieces of 16 B per tile (K 512, V 512), four per thread, two tiles ahead
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    unsigned st_voff[4];
    int st_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + NT * i, is_v = p >= 512, q = p & 511, row = q >> 3, c = q & 7;
        st_voff[i] = unsigned(row) * unsigned(rs) * 2u + unsigned((is_v ? 2 * D : D) + head * HD) * 2u + unsigned(c) * 16u;
        st_lds[i] = is_v ? K_BYTES + row * VST * 2 + c * 16 : row * KST * 2 + c * 16;
    }
    u32x4 sreg[4] = {};
    auto load_tile = [&](int kt) {
        if (kt * KT >= N) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, st_voff[i], kt * tile_bytes, 0);
    };
    auto store_tile = [&](int kt) {
        char* S = lds + (kt % NST) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(S + st_lds[i]) = sreg[i];
    };
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    const int k_off = (r * KST + 8 * h) * 2;
    const int tr_off = ((4 * h + ((lane & 15) >> 2)) * VST + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    f16x8 kreg[4][2], vreg[4][2];
    auto read_k = [&](int st, int kg) __attribute__((always_inline)) {
        const char* p = lds + st * STAGE_BYTES + k_off + 32 * kg;
        kreg[kg][0] = *reinterpret_cast<const f16x8*>(p);
        kreg[kg][1] = *reinterpret_cast<const f16x8*>(p + 32 * KST * 2);
    };
    auto read_v = [&](int st, int g, int dt) __attribute__((always_inline)) {
        const char* p = lds + st * STAGE_BYTES + K_BYTES + tr_off + (16 * g * VST + 32 * dt) * 2;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
        const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 8 * VST * 2));
        vreg[g][dt] = cat(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, c));
    };
    auto wait_k = [&]() {};
    auto wait_v = [&]() {};
    auto fence_k = [&](f32x16&, f32x16&, f32x16&, f32x16&) {};
    auto vcat = [&](const f16x8& f) { return f; };

    f32x16 o[NQ][2], sb[2][NQ][2];
    f32x2 l_run[NQ];
    float m_run[NQ];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[qb][0][i] = 0.f; o[qb][1][i] = 0.f; }
        l_run[qb] = f32x2{0.f, 0.f};
        m_run[qb] = -INFINITY;
    }
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    const int nkt = (N + KT - 1) / KT;

    auto mask_tail = [&](int kt, f32x16& d0, f32x16& d1) {   // padded keys of the last tile
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = kt * KT + mfma32_row(i, h);
            if (key >= N) d0[i] = -INFINITY;
            if (key + 32 >= N) d1[i] = -INFINITY;
        }
    };
    // the tile whose raw scores wait in (n0, n1) joins query block qb's running maximum: o and l are rescaled when a row's maximum
    // moved, and the scores become s - m + 10 (lane maximum `mt` taken beforehand)
    auto join = [&](int qb, float mt, f32x16& n0, f32x16& n1) __attribute__((always_inline)) {
        float ma, mb;
        pope_xor32_pair(mt, ma, mb);                          // the row lives in lanes l and l ^ 32
        const float m_new = __builtin_fmaxf(m_run[qb], __builtin_fmaxf(ma, mb));
        if (__builtin_amdgcn_ballot_w64(m_new > m_run[qb]) != 0) {   // exact: alpha == 1 for the rows that did not move
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);   // first tile: exp2(-inf) = 0 on o = l = 0
            l_run[qb] = l_run[qb] * alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o[qb][0][e] *= alpha; o[qb][1][e] *= alpha; }
        }
        m_run[qb] = m_new;
        const float nshift = 10.0f - m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) { n0[i] += nshift; n1[i] += nshift; }
    };
    auto lane_max = [&](const f32x16& n0, const f32x16& n1) {
        float mt = max3(n0[0], n1[0], n0[1]);
#pragma unroll
        for (int i = 1; i < 15; ++i) mt = max3(mt, n1[i], n0[i + 1]);
        return __builtin_fmaxf(mt, n1[15]);
    };

    // ---- prologue: tiles 0 and 1 in LDS, tile 2 in registers
    load_tile(0);
    store_tile(0);
    load_tile(1);
    if (nkt > 1) store_tile(1);
    load_tile(2);
    if (nkt > 2) store_tile(2);
    __syncthreads();
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) read_k(0, kg);
    wait_k();
    {
        const f32x16 zero = {};
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) {
                sb[0][qb][0] = mfma_f16(kreg[kg][0], qf[qb][kg], kg == 0 ? zero : sb[0][qb][0]);
                sb[0][qb][1] = mfma_f16(kreg[kg][1], qf[qb][kg], kg == 0 ? zero : sb[0][qb][1]);
            }
    }
    fence_k(sb[0][0][0], sb[0][0][1], sb[0][NQ - 1][0], sb[0][NQ - 1][1]);
    if (nkt > 1) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) read_k(1, kg);
    }
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
        if (nkt == 1) mask_tail(0, sb[0][qb][0], sb[0][qb][1]);
        join(qb, lane_max(sb[0][qb][0], sb[0][qb][1]), sb[0][qb][0], sb[0][qb][1]);
    }
    wait_k();

    // ---- one iteration: tile t's shifted scores wait in sb[P]; tile t + 1's scores (if any) are formed in sb[P ^ 1] from kreg.
    // Every MFMA is followed by its slice of VALU work and LDS requests, pinned by scheduling fences.  (has_next / has_next2 stay
    // run-time tests: as template flags — three instantiations per parity — the kernel reaches 256 registers with 38 spills, and
    // an allocator under pressure moves the destinations of the asynchronous asm reads; this form needs 196.)
    int t = 0;
    auto iteration = [&](auto ptag) {
        constexpr int P = decltype(ptag)::value;
        const bool has_next = t + 1 < nkt, has_next2 = t + 2 < nkt;
        __syncthreads();   // tile t + 2 (stored at the end of the previous iteration) is published; everyone has left tile t - 1's stage
        load_tile(t + 3);  // into registers; stored into tile t - 1's stage at the end of this iteration
        const int st_cur = t % NST, st_next2 = (t + 2) % NST;
        f32x2 ls[NQ];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) ls[qb] = f32x2{0.f, 0.f};
        // two probabilities of tile t, in place (neighbours of one tuple)
        auto exp_pair = [&](int qb, int idx) __attribute__((always_inline)) {
            f32x16& c = sb[P][qb][idx < 8 ? 0 : 1];
            const int e = 2 * (idx & 7);
            c[e] = __builtin_amdgcn_exp2f(c[e]);
            c[e + 1] = __builtin_amdgcn_exp2f(c[e + 1]);
            ls[qb] += f32x2{c[e], c[e + 1]};
            asm volatile("" : "+v"(ls[qb]));   // keep the running sum in its slot (the optimiser otherwise sinks the chain behind the MFMAs)
        };
        // ---- phase A: S^T(t + 1) = K(t + 1) . Q^T from kreg for both query blocks (a K fragment feeds two MFMAs), each MFMA
        // followed by two exponential pairs of tile t; every second one by the request of a V^T fragment of tile t
        const f32x16 zero = {};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kg = i >> 1, j = i & 1;
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) {
                if (has_next) sb[P ^ 1][qb][j] = mfma_f16(kreg[kg][j], qf[qb][kg], kg == 0 ? zero : sb[P ^ 1][qb][j]);
                if (qb == 0) read_v(st_cur, i >> 1, i & 1);
                exp_pair(qb, 2 * i);
                exp_pair(qb, 2 * i + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
            if (has_next && t + 2 == nkt) mask_tail(t + 1, sb[P ^ 1][qb][0], sb[P ^ 1][qb][1]);
            l_run[qb] += ls[qb];
        }
        fence_k(sb[P ^ 1][0][0], sb[P ^ 1][0][1], sb[P ^ 1][NQ - 1][0], sb[P ^ 1][NQ - 1][1]);
        wait_v();
        // ---- phase B: O^T += V^T(t) . P^T(t) from vreg (a V fragment feeds two MFMAs); score registers 8s..8s+7 of sub-tile u
        // are the B fragment of k-step (u, s); behind the MFMAs: the lane maxima of tile t + 1's scores and the request of one K
        // fragment pair of tile t + 2
        float mt[NQ];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) mt[qb] = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int u = g >> 1, s = g & 1;
            f16x8 ph[NQ];
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) {
                f32x4 p0, p1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p0[e] = sb[P][qb][u][8 * s + e];
                    p1[e] = sb[P][qb][u][8 * s + 4 + e];
                }
                ph[qb] = cat(__builtin_convertvector(p0, f16x4), __builtin_convertvector(p1, f16x4));
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qb = 0; qb < NQ; ++qb) {
                    o[qb][dt] = mfma_f16(vcat(vreg[g][dt]), ph[qb], o[qb][dt]);
                    if (has_next2 && dt == 0 && qb == 0) read_k(st_next2, g);
                    if (has_next) {
                        const f32x16 &n0 = sb[P ^ 1][qb][0], &n1 = sb[P ^ 1][qb][1];
#pragma unroll
                        for (int q = 2 * dt; q < 2 * dt + 2; ++q) {
                            const int i = 4 * g + q;   // slots 0..15: (n0[0], n1[0], n0[1]), then (mt, n1[i], n0[i + 1]) ..., n1[15] last
                            mt[qb] = i == 0 ? max3(n0[0], n1[0], n0[1]) : i < 15 ? max3(mt[qb], n1[i], n0[i + 1]) : __builtin_fmaxf(mt[qb], n1[15]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (has_next) {
#pragma unroll
            for (int qb = 0; qb < NQ; ++qb) join(qb, mt[qb], sb[P ^ 1][qb][0], sb[P ^ 1][qb][1]);
        }
        if ((t + 3) * KT < N) store_tile(t + 3);
        ++t;
    };
    while (t < nkt) {
        iteration(P0{});
        if (t < nkt) iteration(P1{});
    }
    __syncthreads();   // the stages are free: reuse them for the O^T transpose

    // Normalise (the 2^10 of p' cancels), transpose O^T through LDS, store whole 128-byte head rows (f16, value * 8)
    float* Os = smem + (wave * 32) * OST;
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
        const float l_half = l_run[qb][0] + l_run[qb][1];
        const float inv = 8.0f / (l_half + __shfl_xor(l_half, 32));   // K_PLANES_ACT_SCALE rides on the normalisation
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = o[qb][0][4 * g4 + e] * inv; c[e] = o[qb][1][4 * g4 + e] * inv; }
            *reinterpret_cast<f32x4*>(&Os[r * OST + 8 * g4 + 4 * h]) = a;
            *reinterpret_cast<f32x4*>(&Os[r * OST + 32 + 8 * g4 + 4 * h]) = c;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int lr = (lane >> 4) + 4 * i, c4 = (lane & 15) * 4;
            const int qrow = q0 + (wave * NQ + qb) * 32 + lr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Os[lr * OST + c4]);
            if (qrow < N)
                *reinterpret_cast<f16x4*>(out + (size_t(b) * N + qrow) * D + head * HD + c4) = __builtin_convertvector(v, f16x4);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

static_assert(K_PLANES_ACT_SCALE == 8.0f, "attention f16 epilogue scale");

// qkv: f16 [B * N, 3 * heads * 64] with q pre-scaled (EPI_QKV_F16); out: f16 [B * N, heads * 64], value * 8
int pope_launch_attention_f16_dma(const void* qkv_f16, void* out_f16, int B, int N, int heads, hipStream_t stream) {
    if (!qkv_f16 || !out_f16 || B <= 0 || N <= 0 || heads <= 0 || size_t(B) * heads * ((N + QB - 1) / QB) > 0x7fffffffull) return POPE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(qkv_f16) & 15) || (reinterpret_cast<uintptr_t>(out_f16) & 15)) return POPE_ERR_ARG;
    if (size_t(N + KT) * 3 * heads * HD * 2 >= (size_t(1) << 32) - 512) return POPE_ERR_ARG;
    const dim3 grid(unsigned((N + QB - 1) / QB) * heads * B);
    static pope_dev_mask lds_ok{0};
    if (!pope_opt_in_lds(attn_f16_dma_kernel, F16_ATTN_LDS, lds_ok)) return POPE_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_f16_dma_kernel, grid, dim3(NT), F16_ATTN_LDS, stream, static_cast<const _Float16*>(qkv_f16),
                       static_cast<_Float16*>(out_f16), N, heads);
    return pope_check_launch();
}
