// "f16x3" NT GEMM: fp32 operands split on the fly into hi + lo f16 halves (x = hi + lo with
// hi = f16(x), lo = f16(x - hi): 22 significand bits), three v_mfma_f32_32x32x16_f16 per product
// block (hi.hi + hi.lo + lo.hi, fp32 accumulate).  This moves the contraction from the f32 MFMA —
// which on gfx950 runs on the VALU lanes at 64 FLOP/clk/SIMD — to the real matrix cores
// (1024 FLOP/clk/SIMD for f16): 16/3 = 5.3x the fp32-MFMA rate at ~2^-21 relative accuracy, and
// the VALU work (splitting, epilogues) overlaps with the matrix pipe instead of stealing from it.
//
// Same interface, tiling (128x128x32, 4 waves x 2x2 32x32 tiles), buffer-load register staging,
// LDS-transposed epilogue and epilogue modes as gemm_f32.hip; fp32 in, fp32 out.
// Both operands are multiplied by a power of two before the split (activations 2^3, weights 2^8 by
// default; exact, undone by one exact multiply in the epilogue) so that the lo half of every
// element that matters stays in the f16 normal range: the representation is relative (2^-22) for
// |a| >= 2^-6 and |w| >= 2^-11 and absolute (2^-28 resp. 2^-33) below — far under the fp32 chain's
// own rounding for O(1) activations.  Range contract: |a| < 8188, |w| < 255 (f16 max / scale).
#include <cstdlib>
#include "gemm_core.h"
#include "kernels.h"

namespace {

using namespace gemm_core;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int PL_ST = 40;                       // plane row stride in halves: 80 B = 5 x 16 B (odd) -> b128 reads conflict-free
constexpr int PLANE = 128 * PL_ST;              // halves per plane (128 rows)
constexpr int STAGE = 4 * PLANE;                // A hi, A lo, W hi, W lo
// TWO LDS stages of 40 KB: exactly two workgroups per CU (2 x 80 KB = the CU's 160 KB).  The
// split + ds_write of K-step t+1 goes to the other stage while the MFMAs of K-step t run, so a
// K-step has ONE barrier and its VALU / LDS-write / global-load work sits between its MFMAs (the f16
// MFMA has its own pipe; co-resident waves run in lockstep and would not hide it for each other).
constexpr size_t X3_LDS_BYTES = size_t(2) * STAGE * sizeof(_Float16);
static_assert(X3_LDS_BYTES >= size_t(4) * 32 * EPI_ST * sizeof(float), "epilogue staging must fit");

// Planes kernel: LDS rows keep the memory layout of a K-step, [32 hi | 32 lo] halves + 16 B pad = 144 B.
// The eight lanes that fetch one row's 128 contiguous bytes also write 128 contiguous LDS bytes (all 32
// store banks once: the separate-plane layout above put the hi and lo pieces on the same banks, a 2-way
// conflict on every ds_write_b128 — 8.7 conflict cycles per store in the PMC run), and the 36-dword row
// stride keeps the 16 rows of a ds_read_b128 lane group on 16 distinct 4-bank sets.
constexpr int ROW2 = 72;                        // halves per LDS row
constexpr int OPER2 = 128 * ROW2;               // halves per operand tile (128 rows, both planes)
constexpr int STAGE2 = 2 * OPER2;               // A, W
constexpr size_t X3P_LDS_BYTES = size_t(2) * STAGE2 * sizeof(_Float16);
static_assert(size_t(STAGE2) * sizeof(_Float16) >= size_t(4) * 32 * EPI_ST * sizeof(float), "epilogue staging must fit a stage");

constexpr float A_SCALE = 8.0f, W_SCALE = 256.0f;  // powers of two: exact

__device__ __forceinline__ void split(f32x4 v, float scale, f16x4& hi, f16x4& lo) {
    pope_split4(v * scale, hi, lo);   // common.h: v_cvt_pk_f16_f32 x2 + v_fma_mixlo/mixhi_f16 x2
}

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf_scalar(float x);  // defined below (shared formula with gemm_f32.hip)
// the same formula on a pair (v_pk_fma_f32 / v_pk_mul_f32 for the polynomial): gemm_f32.hip:gelu_erf2
__device__ __forceinline__ f32x2 gelu_erf_pair(f32x2 x) {
    constexpr float P = 0.3275911f * 0.70710678118654752440f;
    constexpr float A1 = 0.5f * 0.254829592f, A2 = 0.5f * -0.284496736f, A3 = 0.5f * 1.421413741f,
                    A4 = 0.5f * -1.453152027f, A5 = 0.5f * 1.061405429f;
    constexpr float NHL2E = -0.5f * 1.44269504088896340736f;
    f32x2 t, e, relu;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        t[i] = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x[i]), P, 1.0f));
        relu[i] = __builtin_fmaxf(x[i], 0.0f);
    }
    const f32x2 arg = (x * NHL2E) * x;
    e[0] = __builtin_amdgcn_exp2f(arg[0]);
    e[1] = __builtin_amdgcn_exp2f(arg[1]);
    f32x2 poly = __builtin_elementwise_fma(t, f32x2{A5, A5}, f32x2{A4, A4});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A3, A3});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A2, A2});
    poly = __builtin_elementwise_fma(poly, t, f32x2{A1, A1});
    const f32x2 q = (poly * t) * e;
    return __builtin_elementwise_fma(relu, __builtin_elementwise_fma(q, f32x2{-2.f, -2.f}, f32x2{1.f, 1.f}), x * q);
}

// LAB is 0 in the product; scripts/x3_lab.cpp times ablations: bit0 = no global loads after the
// prologue, bit1 = no split / LDS writes after the first stage, bit2 = no MFMAs.
template <int EPI, int LAB = 0>
__global__ __launch_bounds__(THREADS, 2) void gemm_nt_f16x3_kernel(const GemmParams g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;

    const BufferLoader la(g.A, g.M, g.lda, m0), lw(g.W, g.N, g.ldw, n0);
    const int nk = g.K / BK;
    // One register staging set.  During K-step t it holds K-step t+1 (split into the other LDS stage)
    // and is then refilled with K-step t+2, which has a whole K-step plus a barrier to arrive.
    f32x4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = la.load(i, 0);
        rw[i] = lw.load(i, 0);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    auto stage_write = [&](int st) {
        _Float16* Ah = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 hi, lo;
            const int o = (srow + 32 * i) * PL_ST + scol;
            split(ra[i], A_SCALE, hi, lo);
            *reinterpret_cast<f16x4*>(Ah + o) = hi;
            *reinterpret_cast<f16x4*>(Ah + PLANE + o) = lo;
            split(rw[i], W_SCALE, hi, lo);
            *reinterpret_cast<f16x4*>(Ah + 2 * PLANE + o) = hi;
            *reinterpret_cast<f16x4*>(Ah + 3 * PLANE + o) = lo;
        }
    };
    const int a_off = (wm * 64 + r) * PL_ST + 8 * h, w_off = 2 * PLANE + (wn * 64 + r) * PL_ST + 8 * h;
    // K-step kt: MFMAs on stage kt&1; meanwhile the staged registers (K-step kt+1) are split into the
    // other stage and then refilled with K-step kt+2.
    // MODE 2: steady state (split + refill), 1: split only (K-step kt+2 does not exist), 0: last K-step.
    // The variants are separate straight-line bodies so that each K-step is ONE basic block and the
    // scheduler can interleave its MFMAs with the split / LDS / load instructions.
    auto kstep = [&](int kt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const _Float16* S = lds + (kt & 1) * STAGE;
        if constexpr (MODE >= 1 && !(LAB & 2)) stage_write((kt + 1) & 1);
        if constexpr (MODE == 2 && !(LAB & 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = la.load(i, (kt + 2) * BK);
                rw[i] = lw.load(i, (kt + 2) * BK);
            }
        }
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            f16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(S + a_off + t * 32 * PL_ST + kg * 16);
                al[t] = *reinterpret_cast<const f16x8*>(S + PLANE + a_off + t * 32 * PL_ST + kg * 16);
                wh[t] = *reinterpret_cast<const f16x8*>(S + w_off + t * 32 * PL_ST + kg * 16);
                wl[t] = *reinterpret_cast<const f16x8*>(S + PLANE + w_off + t * 32 * PL_ST + kg * 16);
            }
            // accumulators hold C^T (rows over n): A-operand = W fragment, B-operand = A fragment
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (LAB & 4) {
                        acc[mi][ni][0] += float(wl[ni][0]) + float(ah[mi][0]) + float(wh[ni][1]) + float(al[mi][1]);
                    } else {
                        acc[mi][ni] = mfma_f16(wl[ni], ah[mi], acc[mi][ni]);  // small terms first
                        acc[mi][ni] = mfma_f16(wh[ni], al[mi], acc[mi][ni]);
                        acc[mi][ni] = mfma_f16(wh[ni], ah[mi], acc[mi][ni]);
                    }
                }
        }
        // 24 MFMAs : ~130 VALU (split) : 16 LDS reads : 16 LDS writes : 8 buffer loads — pin an even mix
        // (LLVM sched groups: 0x8 MFMA, 0x2 VALU, 0x100 DS read, 0x200 DS write, 0x20 VMEM read)
        if constexpr (MODE >= 1 && LAB == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                if (i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i >= 4 && i < 20) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (MODE == 2 && i >= 12 && i < 20) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __syncthreads();  // stage (kt+1)&1 is published, stage kt&1 is free
    };

    stage_write(0);  // K-step 0
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = la.load(i, BK);
            rw[i] = lw.load(i, BK);
        }
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) kstep(kt, std::integral_constant<int, 2>{});
        else if (kt + 1 < nk) kstep(kt, std::integral_constant<int, 1>{});
        else kstep(kt, std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = acc[mi][ni] * (1.0f / (A_SCALE * W_SCALE));

    epilogue_rows(acc, smem, [&](int tr, int tc, f32x4 v) {
        const int row = m0 + tr, col = n0 + tc;
        if (row >= g.M || col >= g.N) return;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias = *reinterpret_cast<const f32x4*>(g.bias + col);
        if constexpr (EPI == EPI_BIAS) {
            v = v + bias;
        } else if constexpr (EPI == EPI_BIAS_GELU) {
            v = v + bias;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf_scalar(v[e]);
        } else {
            const f32x4 gamma = *reinterpret_cast<const f32x4*>(g.gamma + col);
            const f32x4 res = *reinterpret_cast<const f32x4*>(g.res + size_t(row) * g.ldres + col);
            v = res + (v + bias) * gamma;
        }
        *reinterpret_cast<f32x4*>(g.C + size_t(row) * g.ldc + col) = v;
    });
}

// ---- planes kernel -----------------------------------------------------------------------------
// Operands arrive already split (f16 hi/lo planes written once by their producer: LayerNorm, the GELU
// epilogue, the weight loader), so the K loop is loads -> ds_write_b128 -> ds_read_b128 -> MFMA with
// no VALU work at all; the epilogue can emit planes for the next GEMM.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// Persistent: two workgroups per CU walk the tile list (logical id = xcd_remap(blockIdx) + i*grid:
// an XCD keeps whole A row panels in its L2) and treat the K-steps of consecutive tiles as ONE
// stream through the double-buffered LDS: while the last K-step of a tile runs, the first stage of
// the next tile is already being loaded and written, so neither the load latency of a tile's first
// K-steps nor a workgroup re-launch sits between tiles (with K = 384 a tile is only 12 K-steps:
// prologue + epilogue + launch gap were half of a one-tile workgroup's lifetime).
// BMT = 128: 4 waves, two workgroups per CU.  BMT = 256: 8 waves (4 x 2 of 64x64), ONE workgroup per CU whose two
// 128-row halves share the W tile in LDS: 25 % fewer LDS store bytes per MFMA (the kernel is LDS-pipe bound:
// DESIGN.md §4 finding 6) and one barrier domain per CU.
template <int EPI, bool OUT_PLANES, int BMT>
__global__ __launch_bounds__(BMT * 2, 2) void gemm_nt_f16x3_planes_kernel(const GemmParams g, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);
    constexpr int NTH = BMT * 2, SLOTS = NTH / 8;        // threads; rows covered by one pass of 16-byte pieces
    constexpr int NA = BMT / SLOTS, NWR = BN / SLOTS;     // A / W rows per thread and K-step (4 / 4 or 4 / 2)
    constexpr int NLD = NA + NWR;
    constexpr int STAGE_T = (BMT + BN) * ROW2;            // halves per LDS stage: A rows then W rows
    constexpr int OPER_T = BMT * ROW2;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_pb = ((g.M + BMT - 1) / BMT) * tiles_n;  // tiles per batch (EPI_SIM)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    // Staging: a row's K-step is 128 contiguous bytes in memory ([32 hi | 32 lo] halves) = eight 16-byte
    // pieces = eight consecutive lanes -> whole cache lines, and the same 128 contiguous bytes in LDS.
    // Thread -> rows prow + 32 i (i < 4), piece pc.
    const int slot = tid >> 3, pc = tid & 7;
    const int prow = slot;  // rows prow + SLOTS * i
    const int lds_piece = 8 * pc;
    const int nk = g.K / BK;  // >= 2 (launcher)
    const unsigned nb = EPI == EPI_SIM ? unsigned(g.nbatch) : 1u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a_pl), 0, nb * unsigned(g.M) * unsigned(g.lda) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_pl), 0, nb * unsigned(g.N) * unsigned(g.ldw) * 4u, 0x00020000);
    // K-steps of this workgroup's tiles form ONE stream (item = (tile, kt)); two register sets keep
    // the loads of items s+2 and s+3 in flight while item s runs from LDS stage s&1 and item s+1 is
    // written to the other stage: the CU has ~128 KB of loads outstanding, which is what it takes to
    // cover the L2 round trip at this tile size (one set = 64 KB in flight measured ~23 B/clk/CU).
    u32x4 r0[NLD], r1[NLD];  // A rows, then W rows
    // Tile order: static round robin over the XCD-remapped workgroup id.  (The two workgroups of a CU do not
    // progress evenly — the first-dispatched one runs ~1.4x faster and the other finishes its share alone — but
    // neither an atomic tile queue nor an unequal static split shortened the launch: DESIGN.md §4, finding 4.)
    // Full rounds: tile = round * grid + XCD-remapped id.  The last, partial round is dealt by raw blockIdx instead:
    // its tiles then go to workgroups 0, 1, 2, ... = one per CU across all XCDs (each runs alone on its CU, at the
    // solo rate), instead of filling both workgroup slots of the first XCDs' CUs while the other XCDs idle.
    const int grid = gridDim.x, full_rounds = n_tiles / grid;
    const int remapped = xcd_remap(blockIdx.x, grid);
    // The stream bookkeeping must not become control flow: a K-step has to stay ONE basic block for the pinned
    // instruction mix below (with compiler-chosen branches here, the LDS writes and loads of an item form a block of
    // their own in front of the MFMAs and the matrix pipe idles under them) -> asm selects on scalars.
    const int tail_cand = full_rounds * grid + int(blockIdx.x);
    const int tail_tile = tail_cand < n_tiles ? tail_cand : n_tiles;
    auto tile_of = [&](int ord) -> int {
        const int in_tail = pope_uniform_select(ord == full_rounds, tail_tile, n_tiles);
        return pope_uniform_select(ord < full_rounds, ord * grid + remapped, in_tail);
    };
    const int first = tile_of(0);
    if (first >= n_tiles) return;
    int ld_ord = 0, ord = 0;
    int ld_tile = first, ld_kt = 0;  // next stream item to load
    auto load_next = [&](u32x4 (&st)[NLD]) {
        {   // branch-free (an item must be ONE basic block for the interleave below): past the end of the
            // stream the last tile is re-loaded and never consumed
            const int lt = ld_tile < n_tiles ? ld_tile : n_tiles - 1;
            int m0, n0;
            if constexpr (EPI == EPI_SIM) {  // batched: rows of batch b start at b * M (A) / b * N (W)
                const int b = lt / tiles_pb, rem = lt - b * tiles_pb;
                m0 = b * g.M + (rem / tiles_n) * BMT;
                n0 = b * g.N + (rem % tiles_n) * BN;
            } else {
                m0 = (lt / tiles_n) * BMT;
                n0 = (lt % tiles_n) * BN;
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
#ifdef X3_L1ONLY  // timing experiment: every load hits the same few lines (no L2 traffic; wrong results)
                const unsigned va = unsigned((prow + SLOTS * i) & 7) * unsigned(g.lda) * 4u + pc * 16u + 0 * m0;
#else
                const unsigned va = unsigned(m0 + prow + SLOTS * i) * unsigned(g.lda) * 4u + pc * 16u;
#endif
                st[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, va, ld_kt * 128, 0);
            }
#pragma unroll
            for (int i = 0; i < NWR; ++i) {
#ifdef X3_L1ONLY
                const unsigned vw = unsigned((prow + SLOTS * i) & 7) * unsigned(g.ldw) * 4u + pc * 16u + 0 * n0;
#else
                const unsigned vw = unsigned(n0 + prow + SLOTS * i) * unsigned(g.ldw) * 4u + pc * 16u;
#endif
                st[NA + i] = __builtin_amdgcn_raw_buffer_load_b128(rw, vw, ld_kt * 128, 0);
            }
            const int wrap = ++ld_kt == nk;
            ld_kt = pope_uniform_select(wrap, 0, ld_kt);
            ld_ord += wrap;
            ld_tile = tile_of(ld_ord);
        }
    };
    auto write_stage = [&](int s, const u32x4 (&st)[NLD]) {
        _Float16* S = lds + s * STAGE_T + lds_piece;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(S + (prow + SLOTS * i) * ROW2) = st[i];
#pragma unroll
        for (int i = 0; i < NWR; ++i) *reinterpret_cast<u32x4*>(S + OPER_T + (prow + SLOTS * i) * ROW2) = st[NA + i];
    };
    const int a_off = (wm * 64 + r) * ROW2 + 8 * h, w_off = OPER_T + (wn * 64 + r) * ROW2 + 8 * h;
    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;
    };
    struct Frags { f16x8 ah[2], al[2], wh[2], wl[2]; };
    auto read_frags = [&](int s, int kg, Frags& f) {
        const _Float16* S = lds + s * STAGE_T;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f.ah[t] = *reinterpret_cast<const f16x8*>(S + a_off + t * 32 * ROW2 + kg * 16);
            f.al[t] = *reinterpret_cast<const f16x8*>(S + 32 + a_off + t * 32 * ROW2 + kg * 16);
            f.wh[t] = *reinterpret_cast<const f16x8*>(S + w_off + t * 32 * ROW2 + kg * 16);
            f.wl[t] = *reinterpret_cast<const f16x8*>(S + 32 + w_off + t * 32 * ROW2 + kg * 16);
        }
    };
    auto mfma_frags = [&](const Frags& f) {
        // term-major: consecutive MFMAs go to different accumulators (a chain on one accumulator would wait for the
        // previous result); small terms first
#ifdef X3_ACC_MAJOR
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc[mi][ni] = mfma_f16(f.wl[ni], f.ah[mi], acc[mi][ni]);
                acc[mi][ni] = mfma_f16(f.wh[ni], f.al[mi], acc[mi][ni]);
                acc[mi][ni] = mfma_f16(f.wh[ni], f.ah[mi], acc[mi][ni]);
            }
#else
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_f16(f.wl[ni], f.ah[mi], acc[mi][ni]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_f16(f.wh[ni], f.al[mi], acc[mi][ni]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma_f16(f.wh[ni], f.ah[mi], acc[mi][ni]);
#endif
    };
    // Epilogue of one tile, branch-free and free of loads between its stores: bias / gamma are per-lane
    // constants of the tile (a lane keeps its four columns for all 16 row pieces) and are fetched once, up
    // front, under the LDS transposition; rows >= M and columns >= N are dropped by the buffer descriptor's
    // range check instead of exec-mask branches.  (The first version loaded bias inside every piece: the
    // compiler then has to drain vmcnt to 0 before each store — loads and stores share the counter and may
    // retire out of order — so every one of the 16 stores waited for the previous store to reach memory.)
    // The residual variant needs one load per piece: fetched eight at a time, two drain points per tile.
    const unsigned c_row_bytes = unsigned(g.ldc) * 4u;  // fp32 rows and planes rows have the same pitch
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        OUT_PLANES ? g.c_pl : static_cast<void*>(g.C), 0, nb * unsigned(g.M) * c_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == EPI_BIAS_LS_RES ? g.res : g.C), 0,
        EPI == EPI_BIAS_LS_RES ? unsigned(g.res_mod > 0 ? g.res_mod : g.M) * unsigned(g.ldres) * 4u : 0u, 0x00020000);
    auto res_row = [&](unsigned row) -> unsigned { return g.res_mod > 0 ? row % unsigned(g.res_mod) : row; };
    const int ec4 = (lane & 15) * 4, elr = lane >> 4;
    // 8-wave variant: the staging of eight 32x68 wave tiles does not fit one LDS stage, so each wave transposes one
    // 32x32 accumulator block at a time through a 32x36-float region (eight lanes per 128-byte row segment).
    constexpr int EPI_ST2 = 36;
    static_assert(BMT == 128 || size_t(8) * 32 * EPI_ST2 * sizeof(float) <= size_t(STAGE_T) * sizeof(_Float16), "staging fits a stage");
    auto epilogue256 = [&](int tile, float* epi) {
        const int m0 = (tile / tiles_n) * BMT, n0 = (tile % tiles_n) * BN;
        const int c4 = (lane & 7) * 4, lr = lane >> 3;
        constexpr float inv = 1.0f / (A_SCALE * W_SCALE);
        constexpr unsigned DROP = 0xFFFFFF00u;
        f32x4 bias[2], gamma[2];
        bool col_ok[2];
        int col[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            col[ni] = n0 + wn * 64 + ni * 32 + c4;
            col_ok[ni] = col[ni] < g.N;
            const int colc = col_ok[ni] ? col[ni] : 0;
            bias[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            gamma[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g.bias) bias[ni] = *reinterpret_cast<const f32x4*>(g.bias + colc);
            if constexpr (EPI == EPI_BIAS_LS_RES) {
                gamma[ni] = g.gamma ? *reinterpret_cast<const f32x4*>(g.gamma + colc) : f32x4{1.f, 1.f, 1.f, 1.f};
                bias[ni] = bias[ni] * gamma[ni];
                gamma[ni] = gamma[ni] * inv;
            }
        }
        __syncthreads();  // all waves have finished reading the last K-step stage
        float* E = epi + wave * 32 * EPI_ST2;
        f32x2 amax = {0.f, 0.f};  // OUT_PLANES: largest magnitude written as planes (range guard)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g4 + e];
                    *reinterpret_cast<f32x4*>(&E[r * EPI_ST2 + 8 * g4 + 4 * h]) = v;
                }
                const unsigned row0 = unsigned(m0 + wm * 64 + mi * 32 + lr);
                f32x4 res[4];
                if constexpr (EPI == EPI_BIAS_LS_RES) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                     rres, col_ok[ni] ? res_row(row0 + 8 * i) * unsigned(g.ldres) * 4u + unsigned(col[ni]) * 4u : DROP, 0, 0));
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&E[(lr + 8 * i) * EPI_ST2 + c4]);
                    const unsigned off = (row0 + 8 * i) * c_row_bytes;
                    if constexpr (EPI == EPI_BIAS) {
                        v = v * inv + bias[ni];
                    } else if constexpr (EPI == EPI_BIAS_GELU) {
                        v = v * inv + bias[ni];
                        {
                            const f32x2 g01 = gelu_erf_pair(f32x2{v[0], v[1]}), g23 = gelu_erf_pair(f32x2{v[2], v[3]});
                            v = f32x4{g01[0], g01[1], g23[0], g23[1]};
                        }
                    } else {
                        v = res[i] + v * gamma[ni] + bias[ni];
                    }
                    if constexpr (OUT_PLANES) {
                        f16x4 hi, lo;
                        pope_amax4x2(amax, v);
                        split(v, A_SCALE, hi, lo);
                        const unsigned o = col_ok[ni] ? off + unsigned((col[ni] >> 5) * 128 + (col[ni] & 31) * 2) : DROP;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rc, o, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rc, o + 64u, 0, 2);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc,
                                                               col_ok[ni] ? off + unsigned(col[ni]) * 4u : DROP, 0,
                                                               EPI == EPI_BIAS_LS_RES ? 0 : 2);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        if constexpr (OUT_PLANES) pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) * A_SCALE < POPE_F16_OVERFLOW));
        __syncthreads();  // epilogue staging is drained before the stage is written again
    };
    auto epilogue = [&](int tile, float* epi) {
        if constexpr (BMT == 256) {
            epilogue256(tile, epi);
            return;
        }
        if constexpr (EPI == EPI_SIM) {
            // Similarity tile of batch b: sim = acc / divisor_eff (divisor_eff = T * 2^16: the operand scales are exact
            // powers of two), stored to C[b][row][col], plus this wave's partial softmax statistics of the tile — the
            // dual softmax of coarse_matching.py:119 needs max and sum(exp) of every row AND every column of sim, and
            // computing their per-tile pieces here, from registers, replaces two full passes over the L x S matrix.
            // Rows >= M and columns >= N belong to the next batch's operands (or the zero fill): they are set to -inf
            // right after the scaling, so they vanish from every maximum and every sum; their stores are dropped.
            const int b = tile / tiles_pb, rem = tile - b * tiles_pb;
            const int tm = rem / tiles_n, tn = rem - tm * tiles_n;
            const int m0s = tm * BMT, n0s = tn * BN;
            const int cols = n0s + wn * 64 + ec4;
            constexpr unsigned DROPS = 0xFFFFFF00u;
            constexpr float L2E = 1.44269504088896340736f;
            // x / d with a reciprocal and one correction step (q = x r; e = x - d q (exact fma); q += e r): the correctly
            // rounded quotient for all but pathological divisors, 3 instructions instead of the ~10 of a full division
            const float dv = g.divisor_eff, rdiv = g.rdiv;
            const bool edge = m0s + BMT > g.M || n0s + BN > g.N;   // wave-uniform
            // pin the epilogue arithmetic behind the tile-end branch (the compiler otherwise speculates the scaling
            // and the edge selects into every K-step: 300 VALU instructions per 24 MFMAs)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float x = acc[mi][ni][i];
                        float q = x * rdiv;
                        q = __builtin_fmaf(__builtin_fmaf(-dv, q, x), rdiv, q);
                        acc[mi][ni][i] = q;
                    }
            if (edge) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const bool row_out = m0s + wm * 64 + mi * 32 + r >= g.M;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (row_out || n0s + wn * 64 + ni * 32 + mfma32_row(i, h) >= g.N) acc[mi][ni][i] = -INFINITY;
                }
            }
            __syncthreads();
            float* Es = epi + wave * 32 * EPI_ST;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                // ---- row statistics over this wave's 64 columns, in the accumulator layout: lane (r, h) holds 32 of
                // row r's 64 values, lane (r, h ^ 1) the others
                if (g.row_part) {
                    float m = acc[mi][0][0];
#pragma unroll
                    for (int i = 1; i < 16; ++i) m = __builtin_fmaxf(m, acc[mi][0][i]);
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = __builtin_fmaxf(m, acc[mi][1][i]);
                    float ma, mb;
                    pope_xor32_pair(m, ma, mb);
                    m = __builtin_fmaxf(ma, mb);
                    const float ms = m == -INFINITY ? 0.f : m;   // an all-padding block contributes (max -inf, sum 0)
                    f32x2 sum = {0.f, 0.f};
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int i = 0; i < 16; i += 2)
                            sum += f32x2{__builtin_amdgcn_exp2f((acc[mi][ni][i] - ms) * L2E),
                                         __builtin_amdgcn_exp2f((acc[mi][ni][i + 1] - ms) * L2E)};
                    float sa, sb;
                    pope_xor32_pair(sum[0] + sum[1], sa, sb);
                    const int row = m0s + wm * 64 + mi * 32 + r;
                    if (h == 0 && row < g.M)
                        *reinterpret_cast<f32x2*>(g.row_part + ((size_t(b) * g.M + row) * g.ncb + tn * 2 + wn) * 2) = f32x2{m, sa + sb};
                }
                // ---- transposition to rows, coalesced store of sim, column statistics over this block's 32 rows
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g4 + e];
                        *reinterpret_cast<f32x4*>(&Es[r * EPI_ST + ni * 32 + 8 * g4 + 4 * h]) = v;
                    }
                __builtin_amdgcn_wave_barrier();
                f32x4 vr[8];
                f32x4 cm = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 v = vr[i] = *reinterpret_cast<const f32x4*>(&Es[(elr + 4 * i) * EPI_ST + ec4]);
                    const int row = m0s + wm * 64 + mi * 32 + elr + 4 * i;
#pragma unroll
                    for (int e = 0; e < 4; ++e) cm[e] = __builtin_fmaxf(cm[e], v[e]);
                    const unsigned off = (unsigned(b) * unsigned(g.M) + unsigned(row)) * c_row_bytes + unsigned(cols) * 4u;
                    const bool row_ok = row < g.M;
                    if (!(g.N & 1)) {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v[0], v[1]}), rc,
                                                              row_ok && cols + 1 < g.N ? off : DROPS, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v[2], v[3]}), rc,
                                                              row_ok && cols + 3 < g.N ? off + 8u : DROPS, 0, 2);
                    } else {  // odd row length: rows are only 4-byte aligned, plain element stores
                        float* cp = g.C + (size_t(b) * g.M + row) * g.N + cols;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row_ok && cols + e < g.N) __builtin_nontemporal_store(v[e], cp + e);
                    }
                }
                if (g.col_pmax) {
                    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // the four lane groups (lane >> 4) hold 8 rows each
                        float a, c;
                        pope_xor16_pair(cm[e], a, c);
                        pope_xor32_pair(__builtin_fmaxf(a, c), a, c);
                        cm[e] = __builtin_fmaxf(a, c);
                    }
                    f32x4 cms;
#pragma unroll
                    for (int e = 0; e < 4; ++e) cms[e] = cm[e] == -INFINITY ? 0.f : cm[e];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) cs[e] += __builtin_amdgcn_exp2f((vr[i][e] - cms[e]) * L2E);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a, c;
                        pope_xor16_pair(cs[e], a, c);
                        pope_xor32_pair(a + c, a, c);
                        cs[e] = a + c;
                    }
                    if (elr == 0 && cols < g.N) {   // ldp is a multiple of 4 and cols too: the quad never leaves the row
                        const size_t o = (size_t(b) * g.nrb + tm * 4 + wm * 2 + mi) * g.ldp + cols;
                        *reinterpret_cast<f32x4*>(g.col_pmax + o) = cm;
                        *reinterpret_cast<f32x4*>(g.col_psum + o) = cs;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            return;
        }
#ifdef X3_NO_EPILOGUE  // dev timing floor (wrong results)
        {
            float sum = 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum += acc[mi][ni][e];
            if (sum == 1234.5f) g.C[0] = 1.f;
            __syncthreads();
        }
        return;
#endif
        const int m0 = (tile / tiles_n) * BMT, n0 = (tile % tiles_n) * BN;
        const int col = n0 + wn * 64 + ec4;
        const bool col_ok = col < g.N;
        const int colc = col_ok ? col : 0;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f}, gamma = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias = *reinterpret_cast<const f32x4*>(g.bias + colc);
        if constexpr (EPI == EPI_BIAS_LS_RES) gamma = g.gamma ? *reinterpret_cast<const f32x4*>(g.gamma + colc) : f32x4{1.f, 1.f, 1.f, 1.f};
        constexpr float inv = 1.0f / (A_SCALE * W_SCALE);
        if constexpr (EPI == EPI_BIAS_LS_RES) {  // res + (v*inv + bias)*gamma = res + v*(inv*gamma) + bias*gamma
            bias = bias * gamma;
            gamma = gamma * inv;
        }
        constexpr unsigned DROP = 0xFFFFFF00u;  // beyond every buffer extent: the access is discarded
        __syncthreads();  // all waves have finished reading the last K-step stage
        float* E = epi + wave * 32 * EPI_ST;
        f32x2 amax = {0.f, 0.f};  // OUT_PLANES: largest magnitude written as planes (range guard; rows >= M hold finite junk
                           // computed from zero-filled operands: bias / gelu(bias), the same values as real rows see)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g4 + e];
                    *reinterpret_cast<f32x4*>(&E[r * EPI_ST + ni * 32 + 8 * g4 + 4 * h]) = v;
                }
            const unsigned row0 = unsigned(m0 + wm * 64 + mi * 32 + elr);
            f32x4 res[8];
            if constexpr (EPI == EPI_BIAS_LS_RES) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 rres, col_ok ? res_row(row0 + 4 * i) * unsigned(g.ldres) * 4u + unsigned(col) * 4u : DROP, 0, 0));
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&E[(elr + 4 * i) * EPI_ST + ec4]);
                const unsigned off = (row0 + 4 * i) * c_row_bytes;
                if constexpr (EPI == EPI_BIAS) {
                    v = v * inv + bias;
                } else if constexpr (EPI == EPI_BIAS_GELU) {
                    v = v * inv + bias;
                    {
                        const f32x2 g01 = gelu_erf_pair(f32x2{v[0], v[1]}), g23 = gelu_erf_pair(f32x2{v[2], v[3]});
                        v = f32x4{g01[0], g01[1], g23[0], g23[1]};
                    }
                } else {
                    v = res[i] + v * gamma + bias;
                }
                if constexpr (OUT_PLANES) {
                    f16x4 hi, lo;
                    pope_amax4x2(amax, v);
                    split(v, A_SCALE, hi, lo);
                    // planes row: per 32-column chunk [32 hi | 32 lo] halves.  (Trading halves between neighbouring
                    // lanes so that each lane issues one 16-byte store — even lanes hi, odd lanes lo — was 6 % slower.)
                    const unsigned o = col_ok ? off + unsigned((col >> 5) * 128 + (col & 31) * 2) : DROP;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rc, o, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rc, o + 64u, 0, 2);
                } else {
                    // write-once outputs (qkv: 450 MB per launch, far beyond L2) are stored non-temporally so
                    // they do not displace the A/W panels in L2 (+5 %); the residual stream (LS_RES) is re-read
                    // by the next LayerNorm and keeps the default policy
#ifdef X3_NOSTORE  // dev timing experiment: every store is out of range (dropped by the descriptor)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, v[0] == 1234.5f ? 0u : DROP, 0, 0);
#else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc,
                                                           col_ok ? off + unsigned(col) * 4u : DROP, 0,
                                                           EPI == EPI_BIAS_LS_RES ? 0 : 2);
#endif
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (OUT_PLANES) pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) * A_SCALE < POPE_F16_OVERFLOW));
        __syncthreads();  // epilogue staging is drained before the stage is written again
    };

    // prologue: item 0 -> LDS stage 0; items 1, 2 in flight in r1, r0
    load_next(r0);
    write_stage(0, r0);
    load_next(r1);
    load_next(r0);
    __syncthreads();
    zero_acc();
    int tile = first, kt = 0;

    // one stream item; `nx` holds item s+1 (to be published), then is refilled with item s+3
    // Order inside an item: the fragment reads of this item are issued FIRST, their LDS latency runs
    // under the ds_writes / buffer loads of the staging work; the second half's fragments are read
    // while the first half's 12 MFMAs execute (the two waves of a SIMD run in lockstep, so an LDS
    // wait of one is not covered by MFMAs of the other).
#ifdef X3_STAMPS  // dev: cycle stamps of (block, tile ordinal, K-step) into g.posb (scripts/x3_lab.cpp)
    int tile_ord = 0;
    auto stamp = [&](int slot) {
        if (tid == 0 && g.posb && tile_ord < 16 && blockIdx.x < 512)
            reinterpret_cast<unsigned long long*>(const_cast<float*>(g.posb))[(blockIdx.x * 16 + tile_ord) * 16 + slot] =
                __builtin_readcyclecounter();
        if (tid == 0 && g.posb && tile_ord < 16 && blockIdx.x < 512 && (slot == 0 || slot == 13))  // 100 MHz wall clock
            reinterpret_cast<unsigned long long*>(const_cast<float*>(g.posb))[(blockIdx.x * 16 + tile_ord) * 16 + (slot ? 15 : 14)] =
                __builtin_amdgcn_s_memrealtime();
    };
#else
    auto stamp = [&](int) {};
#endif
#ifdef X3_NO_FRAGREAD
    Frags fr0, fr1;
#endif
    auto item = [&](int s, u32x4 (&nx)[NLD]) {
        stamp(kt);
        Frags f0, f1;
#ifdef X3_NO_FRAGREAD  // dev timing experiment (wrong results): fragments are read for the first item only
        if (s < 1) { read_frags(0, 0, fr0); read_frags(0, 1, fr1); }
        f0 = fr0; f1 = fr1;
#else
        read_frags(s & 1, 0, f0);
#endif
#if defined(X3_NO_DSWRITE)   // dev timing experiments (wrong results): loads only / LDS stores only / neither
        load_next(nx);
        asm volatile("" :: "v"(nx[0]), "v"(nx[NLD - 1]));
#elif defined(X3_NO_LOAD)
        write_stage((s + 1) & 1, nx);
        if (++ld_kt == nk) { ld_kt = 0; ld_tile = tile_of(++ld_ord); }
#elif !defined(X3_NO_STAGE)
        write_stage((s + 1) & 1, nx);
        load_next(nx);
#else
        if (++ld_kt == nk) { ld_kt = 0; ld_tile = tile_of(++ld_ord); }
#endif
#ifndef X3_NO_FRAGREAD
        read_frags(s & 1, 1, f1);
#endif
        mfma_frags(f0);
        mfma_frags(f1);
#ifndef X3_NO_SCHED
        // Pin the instruction mix (LLVM sched groups 0x8 MFMA, 0x100 DS read, 0x200 DS write, 0x20 VMEM
        // read): the LDS writes, buffer loads and second-half fragment reads are spread between the 24
        // MFMAs instead of forming their own phases — all waves of a CU run in lockstep, so a phase that
        // uses only the LDS or only the load path leaves the matrix pipe idle on the whole CU.
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // f0
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NLD) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (i >= 2 && i < 10) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // f1
            if (i >= 8 && i < 8 + NLD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#endif
        __syncthreads();  // stage (s+1)&1 is published, stage s&1 is free
        if (++kt == nk) {
            stamp(12);
            epilogue(tile, reinterpret_cast<float*>(lds + (s & 1) * STAGE_T));
            stamp(13);
#ifdef X3_STAMPS
            ++tile_ord;
#endif
            zero_acc();
            kt = 0;
            tile = tile_of(++ord);

        }
    };
    for (int s = 0; tile < n_tiles; s += 2) {
        item(s, r1);
        if (tile < n_tiles) item(s + 1, r0);
    }
}

// exact-erf GELU, Abramowitz-Stegun 7.1.26 form (see gemm_f32.hip:gelu_erf2 for the derivation)
__device__ __forceinline__ float gelu_erf_scalar(float x) {
    constexpr float P = 0.3275911f * 0.70710678118654752440f;
    constexpr float A1 = 0.5f * 0.254829592f, A2 = 0.5f * -0.284496736f, A3 = 0.5f * 1.421413741f,
                    A4 = 0.5f * -1.453152027f, A5 = 0.5f * 1.061405429f;
    constexpr float NHL2E = -0.5f * 1.44269504088896340736f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x), P, 1.0f));
    const float e = __builtin_amdgcn_exp2f((x * NHL2E) * x);
    float poly = __builtin_fmaf(t, A5, A4);
    poly = __builtin_fmaf(poly, t, A3);
    poly = __builtin_fmaf(poly, t, A2);
    poly = __builtin_fmaf(poly, t, A1);
    const float q = (poly * t) * e;
    return __builtin_fmaf(__builtin_fmaxf(x, 0.0f), __builtin_fmaf(q, -2.f, 1.f), x * q);
}

template <int EPI>
int launch(const GemmParams& g, hipStream_t stream) {
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_nt_f16x3_kernel<EPI>, X3_LDS_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_nt_f16x3_kernel<EPI>), dim3(tiles), dim3(THREADS), X3_LDS_BYTES, stream, g);
    return pope_check_launch();
}

}  // namespace

// lab entry (not part of the C ABI): launch an ablated variant
int pope_lab_gemm_f16x3(const GemmParams& g, int lab, hipStream_t stream) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
#define POPE_LAB_CASE(L)                                                                                          \
    case L:                                                                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f16x3_kernel<EPI_BIAS, L>),                   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, int(X3_LDS_BYTES));                       \
        hipLaunchKernelGGL((gemm_nt_f16x3_kernel<EPI_BIAS, L>), dim3(tiles), dim3(THREADS), X3_LDS_BYTES, stream, g); \
        break;
    switch (lab) { POPE_LAB_CASE(1) POPE_LAB_CASE(2) POPE_LAB_CASE(3) POPE_LAB_CASE(4) POPE_LAB_CASE(5) POPE_LAB_CASE(6) default: return POPE_ERR_ARG; }
#undef POPE_LAB_CASE
    return pope_check_launch();
}

template <int EPI, bool OUT_PLANES, int BMT>
int launch_planes_t(const GemmParams& g, hipStream_t stream, int nbatch = 1) {
    constexpr size_t lds = size_t(2) * (BMT + BN) * ROW2 * sizeof(_Float16);
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_nt_f16x3_planes_kernel<EPI, OUT_PLANES, BMT>, lds, lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = nbatch * ((g.M + BMT - 1) / BMT) * ((g.N + BN - 1) / BN);
    // 128-row tiles: two resident workgroups per CU (2 x 72 KB LDS); 256-row tiles: one (108 KB)
    const int slots = (BMT == 128 ? 2 : 1) * pope_cu_count();
    hipLaunchKernelGGL((gemm_nt_f16x3_planes_kernel<EPI, OUT_PLANES, BMT>), dim3(tiles < slots ? tiles : slots), dim3(BMT * 2),
                       lds, stream, g, tiles);
    return pope_check_launch();
}

// dev switch for same-box A/B runs: POPE_GEMM_MFMA=32 selects this file's v_mfma_f32_32x32x16_f16 mainloop instead of the
// 16x16x32 kernel of gemm_planes.hip (the default: +13 % executed FLOP/s at the clock the chip holds, see that file)
static bool use_mfma32() {
    static const bool v = getenv("POPE_GEMM_MFMA") && atoi(getenv("POPE_GEMM_MFMA")) == 32;
    return v;
}

template <int EPI, bool OUT_PLANES>
int launch_planes(const GemmParams& g, hipStream_t stream) {
    if (g.plain || !use_mfma32()) return pope_launch_planes16(g, stream);   // the single-product mode lives in gemm_planes.hip only
    static const int force = getenv("POPE_GEMM_BM") ? atoi(getenv("POPE_GEMM_BM")) : 0;  // dev switch: 128 or 256
    // Measured (DESIGN.md §4 finding 6): the 256-row kernel moves 25 % fewer LDS store bytes per MFMA and is 2-5 %
    // faster on QKV / FC1 in isolation, but inside the model (planes outputs, neighbours' cache state) the 128-row
    // kernel, whose second workgroup runs its K loop under the first one's epilogue, wins on every shape: default.
    const bool big = force == 256;
    return big ? launch_planes_t<EPI, OUT_PLANES, 256>(g, stream) : launch_planes_t<EPI, OUT_PLANES, 128>(g, stream);
}

// Batched similarity for the dense matcher: C[b] = (A[b] . W[b]^T * alpha) / divisor on planes operands.
int pope_launch_sim_f16x3_planes(const GemmParams& g, hipStream_t stream) {
    if (g.epilogue != EPI_SIM || !g.a_pl || !g.w_pl || !g.C || g.nbatch <= 0 || g.M <= 0 || g.N <= 0) return POPE_ERR_ARG;
    if (g.K < 2 * BK || (g.K % BK) || (g.lda & 31) || (g.ldw & 31) || g.ldc != g.N || g.divisor_eff == 0.f) return POPE_ERR_ARG;
    if ((g.row_part || g.col_pmax) &&
        (!g.row_part || !g.col_pmax || !g.col_psum || g.ncb != 2 * ((g.N + BN - 1) / BN) || g.nrb != 4 * ((g.M + BM - 1) / BM) ||
         g.ldp < g.N || (g.ldp & 3)))
        return POPE_ERR_ARG;
    if ((size_t(g.nbatch) * g.M + BM) * g.lda * 4 >= (size_t(1) << 32) || (size_t(g.nbatch) * g.N + BN) * g.ldw * 4 >= (size_t(1) << 32) ||
        (size_t(g.nbatch) * g.M + BM) * g.ldc * 4 >= (size_t(1) << 32) - 512)
        return POPE_ERR_ARG;
    if (static_cast<long long>(g.nbatch) * ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) > 0x7fffffffLL) return POPE_ERR_ARG;
    if (!use_mfma32()) return pope_launch_planes16(g, stream);
    return launch_planes_t<EPI_SIM, false, 128>(g, stream, g.nbatch);
}

int pope_launch_gemm_nt_f16x3_planes(const GemmParams& g, hipStream_t stream) {
    static_assert(A_SCALE == K_PLANES_ACT_SCALE && W_SCALE == K_PLANES_W_SCALE, "plane scales");
    if (g.M <= 0 || g.N <= 0 || g.K < 2 * BK || (g.K % BK) || (g.N & 3) || (g.ldc & 3) || (g.lda & 7) || (g.ldw & 7)) return POPE_ERR_ARG;
    if (!g.a_pl || !g.w_pl || (g.lda & 31) || (g.ldw & 31)) return POPE_ERR_ARG;
    if (size_t(g.M + 256) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N + BN) * g.ldw * 4 >= (size_t(1) << 32)) return POPE_ERR_ARG;
    const bool out_planes = g.c_pl != nullptr;
    if (out_planes ? (g.ldc & 31) != 0 : !g.C) return POPE_ERR_ARG;
    // the epilogue addresses C (and res) through 32-bit buffer offsets
    if (size_t(g.M + 256) * g.ldc * 4 >= (size_t(1) << 32) - 512) return POPE_ERR_ARG;
    if (g.epilogue == EPI_BIAS_LS_RES && size_t(g.M + 256) * g.ldres * 4 >= (size_t(1) << 32) - 512) return POPE_ERR_ARG;
#ifdef POPE_XSTAT_LAB   // lab builds only (scripts/xstat_ab.sh): the X-stationary mainloop of scripts/gemm_xstat_lab.hip for QKV / FC1
    if (pope_xstat_supported(g)) return pope_launch_xstat(g, stream);
#endif
    switch (g.epilogue) {
        case EPI_BIAS: return out_planes ? launch_planes<EPI_BIAS, true>(g, stream) : launch_planes<EPI_BIAS, false>(g, stream);
        case EPI_BIAS_GELU:
            return out_planes ? launch_planes<EPI_BIAS_GELU, true>(g, stream) : launch_planes<EPI_BIAS_GELU, false>(g, stream);
        case EPI_BIAS_LS_RES:
            if (!g.res || out_planes || (!g.gamma && g.res_mod <= 0)) return POPE_ERR_ARG;
            return launch_planes<EPI_BIAS_LS_RES, false>(g, stream);
        case EPI_BIAS_RELU: return pope_launch_planes16(g, stream);   // 16x16x32 kernel only
    }
    return POPE_ERR_ARG;
}

bool pope_gemm_f16x3_supported(const GemmParams& g) {
    return g.epilogue != EPI_POSB && (g.K % BK) == 0 && size_t(g.M + BM) * g.lda * 4 < (size_t(1) << 32) &&
           size_t(g.N + BN) * g.ldw * 4 < (size_t(1) << 32);
}

int pope_launch_gemm_nt_f16x3(const GemmParams& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.N & 3) || (g.ldc & 3) || !pope_gemm_f16x3_supported(g)) return POPE_ERR_ARG;
    if (g.range_flag) {  // this kernel splits fp32 operands inside its K loop: check them in a scan of their own
        if (g.lda != g.K || g.ldw != g.K) return POPE_ERR_ARG;
        int rc = pope_launch_range_check(g.A, size_t(g.M) * g.K, A_SCALE, g.range_flag, POPE_RANGE_INPUT, stream);
        if (!rc) rc = pope_launch_range_check(g.W, size_t(g.N) * g.K, W_SCALE, g.range_flag, POPE_RANGE_INPUT, stream);
        if (rc) return rc;
    }
    if ((g.lda & 3) || (g.ldw & 3) || (reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.W) & 15) ||
        (reinterpret_cast<uintptr_t>(g.C) & 15))
        return POPE_ERR_ARG;
    switch (g.epilogue) {
        case EPI_BIAS: return launch<EPI_BIAS>(g, stream);
        case EPI_BIAS_GELU: return launch<EPI_BIAS_GELU>(g, stream);
        case EPI_BIAS_LS_RES:
            if (!g.gamma || !g.res) return POPE_ERR_ARG;
            return launch<EPI_BIAS_LS_RES>(g, stream);
    }
    return POPE_ERR_ARG;
}
