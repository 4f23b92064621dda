// Multi-head attention in single-product f16 arithmetic (POPE_PREC_F16: BASELINE config 5's stated dtype; never the headline
// path).  attention.py:49-62 on operands the QKV GEMM's epilogue has already rounded to f16 (EPI_QKV_F16: q pre-scaled by
// head_dim^-0.5 * log2 e in fp32, then ONE rounding; k, v as they are): qkv[B * N, 3 * heads * 64] f16 -> out[B * N, heads * 64]
// f16 (value * 8, the plain proj GEMM's operand).  fp32 scores, softmax, accumulators and normalisation; P is rounded to f16
// for the second product (what the reference shows under .half()).
//
// Round 4.  Until now this mode ran the single-stage kernel of attention_f16x3.hip on the fp32 output of the QKV GEMM (converted
// per tile on the way into LDS, two barriers per tile).  Here:
//   * K / V rows of a (key, head) are exactly one 128-byte line of the f16 qkv tensor: they go memory -> LDS directly
//     (buffer_load ... lds; no registers, no conversion, no ds_write), into the padded stage layout of the f16x3 kernel (K rows
//     144 B, V rows 192 B: conflict-free fragment and transposing reads); the padding positions fetch from beyond the
//     descriptor's extent (zeros, no traffic); 8 consecutive lanes fetch one whole line; four stages of 21 KB, one raw
//     s_barrier per tile behind an explicit s_waitcnt (a fenced barrier, and the compiler's own handling of the transposing
//     read builtin, wait for EVERY LDS-direct load in flight);
//   * same orientation as the f16x3 kernel: S^T = K . Q^T (a lane holds 16 keys of one query), O^T += V^T . P^T with the score
//     registers converted in place into the B operand, V^T fragments by ds_read_b64_tr_b16;
//   * software pipeline inside each wave: the score MFMAs of tile t + 1 are interleaved with the exponentials of tile t, the P.V
//     MFMAs of tile t with the row maximum of tile t + 1; all fragment reads one phase ahead of their MFMAs; exact per-tile
//     running maximum (p' = 2^(s - m + 10) <= 1 024 always fits f16; no overflow path).
// Measured (ViT-L/14, 21 images x 16 heads x 1 531 tokens): 0.317 ms = 0.25 of the f16 peak, the same as the kernel it replaces —
// what it saves is the fp32 qkv tensor (the QKV GEMM writes half the bytes).  Timing-only ablations show why no schedule helps:
// the parts add up without any overlap (LDS reads + barrier 0.11 ms, score MFMAs 0.08, P.V MFMAs 0.06, exponentials 0.03,
// maximum / rescale 0.03): with ONE MFMA per product a wave reads the whole K and V tile (16 KB) for 16 MFMAs — 128 KB per tile
// and CU, ~1 500 LDS cycles against 1 024 MFMA cycles per SIMD — and the eight waves run their phases in lock step behind the
// per-tile barrier.  The way out is 64 queries per wave (NQ = 2), which does not fit 256 registers next to two score buffers.
#include "common.h"
#include "kernels.h"
#include <type_traits>

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int HD = 64, KT = 64;
constexpr int WAVES = 8, NQ = 1;                        // waves per workgroup, 32-query blocks per wave
// NQ = 2 with four waves (one per SIMD, 512 registers: every K / V fragment feeds two MFMAs, half the LDS reads) was built and
// measured: 0.39 ms against 0.32 at the ViT-L/14 shape — and wrong: at that register pressure the allocator moves the outputs of
// the asynchronous inline-asm LDS reads (VGPR -> AGPR copies) before their data has arrived.  Asm reads are safe only while
// their destinations stay put, i.e. well below the register limit.
static_assert(NQ == 1 && WAVES == 8, "see above");
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

__global__ __launch_bounds__(NT, 2) void attn_f16_dma_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, int N, int heads) {
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
    unsigned dma_voff[NDMA];
    int dma_lds[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        int blk = wave + WAVES * i;
        blk = blk >= KBLK + VBLK ? blk - (KBLK + VBLK) : blk;   // (same bytes to the same place once more)
        const bool is_v = blk >= KBLK;
        const int jb = is_v ? blk - KBLK : blk, ppr = is_v ? 12 : 9;
        const int pidx = jb * 64 + lane, prow = pidx / ppr, c = pidx - prow * ppr;
        dma_lds[i] = (is_v ? K_BYTES : 0) + jb * 1024;
        dma_voff[i] = c < 8 ? unsigned(prow) * unsigned(rs) * 2u + unsigned((is_v ? 2 * D : D) + head * HD) * 2u + unsigned(c) * 16u : 0xFFFFFF00u;
    }
    auto dma_tile = [&](int kt) {
        if (kt * KT >= N) return;   // never address a tile past the last one (the range check subtracts the scalar offset from the extent)
        char* S = lds + (kt % NST) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < NDMA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(S + dma_lds[i]), 16, dma_voff[i], kt * tile_bytes, 0, 0);
    };

    // ---- fragments.  ALL LDS reads of the loop are inline asm with explicit waits: behind the builtin the compiler cannot tell the
    // transposing read from the LDS-direct loads' destinations and puts s_waitcnt vmcnt(0) in front of every one of them (every
    // tile in flight waited for, in every iteration).  Full register sets, read one phase AHEAD: the V^T fragments of tile t
    // under the score MFMAs of tile t + 1, the K fragments of tile t + 2 under the P.V MFMAs of tile t.
    const unsigned lds_base = unsigned(size_t((lds_void_ptr)lds));
    const unsigned k_addr = lds_base + unsigned(r * KST + 8 * h) * 2u;
    // ds_read_b64_tr_b16 (attention_f16x3.hip): within a 16-lane group, lane 4q + p supplies row q, columns 4p..4p+3 of a 4-key x
    // 16-d block and lane i receives column i (its d) of the 4 keys
    const unsigned tr_addr = lds_base + unsigned((4 * h + ((lane & 15) >> 2)) * VST + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2u;
    f16x8 kreg[4][2];                 // K fragments of the tile whose scores are formed next: [kg][key half]
    struct VFrag { s16x4 a, c; };
    VFrag vreg[4][2];                 // V^T fragments of the current tile: [16-key group][d half]
    auto read_k = [&](int st, int kg) __attribute__((always_inline)) {
        const unsigned addr = k_addr + unsigned(st) * STAGE_BYTES;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kreg[kg][0]) : "v"(addr), "i"(32 * kg) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kreg[kg][1]) : "v"(addr), "i"(32 * kg + 32 * KST * 2) : "memory");
    };
    auto read_v = [&](int st, int g, int dt) __attribute__((always_inline)) {
        const unsigned addr = tr_addr + unsigned(st) * STAGE_BYTES;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vreg[g][dt].a) : "v"(addr), "i"(K_BYTES + (16 * g * VST + 32 * dt) * 2) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vreg[g][dt].c) : "v"(addr), "i"(K_BYTES + ((16 * g + 8) * VST + 32 * dt) * 2) : "memory");
    };
    // s_waitcnt lgkmcnt(0) tied to the registers the reads fill: it stays between the reads and the MFMAs that consume them
    auto wait_k = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(kreg[0][0]), "+v"(kreg[0][1]), "+v"(kreg[1][0]), "+v"(kreg[1][1]), "+v"(kreg[2][0]), "+v"(kreg[2][1]), "+v"(kreg[3][0]), "+v"(kreg[3][1])
                     :: "memory");
    };
    auto wait_v = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vreg[0][0].a), "+v"(vreg[0][0].c), "+v"(vreg[0][1].a), "+v"(vreg[0][1].c), "+v"(vreg[1][0].a), "+v"(vreg[1][0].c),
                       "+v"(vreg[1][1].a), "+v"(vreg[1][1].c), "+v"(vreg[2][0].a), "+v"(vreg[2][0].c), "+v"(vreg[2][1].a), "+v"(vreg[2][1].c),
                       "+v"(vreg[3][0].a), "+v"(vreg[3][0].c), "+v"(vreg[3][1].a), "+v"(vreg[3][1].c)
                     :: "memory");
    };
    // the MFMAs that read kreg are ordered in front of the asm reads that refill it
    auto fence_k = [&](f32x16& a, f32x16& b2, f32x16& c, f32x16& d) __attribute__((always_inline)) {
        asm volatile("" : "+v"(a), "+v"(b2));
        asm volatile("" : "+v"(c), "+v"(d), "+v"(kreg[0][0]), "+v"(kreg[1][0]), "+v"(kreg[2][0]), "+v"(kreg[3][0]), "+v"(kreg[0][1]),
                     "+v"(kreg[1][1]), "+v"(kreg[2][1]), "+v"(kreg[3][1]));
    };
    auto vcat = [&](const VFrag& f) { return cat(__builtin_bit_cast(f16x4, f.a), __builtin_bit_cast(f16x4, f.c)); };

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

    // ---- prologue: tiles 0, 1, 2 land; S^T(0); the K fragments of tile 1
    dma_tile(0);
    dma_tile(1);
    dma_tile(2);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    raw_barrier();
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
        // every tile issued so far (<= t + 2) has landed — tile t + 2's K rows are read in this iteration — and everyone has left
        // tile t - 1's stage
        __builtin_amdgcn_s_waitcnt(0x0F70);
        raw_barrier();
        dma_tile(t + 3);                  // one iteration to land
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
        wait_k();
        ++t;
    };
    while (t < nkt) {
        iteration(P0{});
        if (t < nkt) iteration(P1{});
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
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
