// Residual GEMM with the FOLLOWING LayerNorm fused into its epilogue (north star: "fused LayerNorm + projection"):
//     x  <- res + gamma * (A . W^T + bias)                       block.py:105-106 (proj / fc2); patch embed (res = pos table)
//     xn <- LayerNorm(x) * ln_w + ln_b   (eps inside the sqrt)   block.py:56,68 (norm2 / next block's norm1), final norm
// for N = 384 (ViT-S/14).  x is written as fp32 (the residual stream), xn as f16x3 activation planes for the next
// GEMM (or as fp32 for the final norm): the stand-alone LayerNorm launches between the GEMMs — 100 per step, each a
// read of x and a write of the planes at the HBM copy ceiling — disappear; x is normalised while its tile is still
// in the producing workgroup's registers.
//
// A LayerNorm row needs all 384 columns, so the tile is a FULL ROW BLOCK: 128 rows x 384 columns, one workgroup of
// 8 waves per CU (wave (wm, wn) owns 64 rows x 96 columns = 4 x 6 accumulator blocks of v_mfma_f32_16x16x32_f16).
// Against the 128 x 128 kernel's two workgroups per CU this moves a third fewer bytes into LDS per MFMA (the W tile
// is shared by twice the rows) and has one barrier domain; the 766 row tiles of a 64-image chunk are 2.99 rounds of
// 256 CUs.  Operands are planes (gemm_planes.hip: the K-steps of consecutive tiles form one stream through a
// double-buffered LDS, the next K-step's loads in flight in registers); LDS rows are the memory rows (128 B) with the
// 16-byte chunk index XOR-ed with row & 7: conflict-free stores and 16-row fragment reads without padding, 2 x 64 KB.
//
// Epilogue, all in the accumulator layout (lane = row l & 15 of a block, four consecutive columns per register quad):
// residual rows arrive as 16-byte loads and x replaces the accumulators in place; row sums go through two lane swaps
// and a 4 KB LDS table across the four column waves; mean first, then the centred second moment (the two-pass form of
// the stand-alone kernel: no E[x^2] - mean^2 cancellation); x and xn are stored with no load between the stores.
// Reduction orders depend on the column only: batch invariance and run-to-run determinism are kept.
#include "gemm_core.h"
#include "kernels.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int RM = 128, RN = 384, RK = 32, RTH = 512;
constexpr int NMI = 4;   // 16-row accumulator blocks per wave
constexpr int ROWB = 64;                          // halves per LDS row: 128 B, chunk-swizzled
constexpr int STAGE_H = (RM + RN) * ROWB;          // halves per stage (A rows, then W rows): 64 KB
constexpr size_t STATS_OFF = size_t(2) * STAGE_H * sizeof(_Float16);
constexpr size_t CTAB_OFF = STATS_OFF + size_t(2) * 4 * RM * sizeof(float);       // after the two [4][128] row-sum tables
constexpr size_t PF_OFF = CTAB_OFF + size_t(4) * RN * sizeof(float);              // + per-column constants
constexpr size_t RL_LDS_BYTES = PF_OFF + size_t(8) * 256;                         // + landing pad of the residual prefetch = 143 360 B
typedef __attribute__((address_space(3))) void* lds_void_ptr;
constexpr float A_SCALE = K_PLANES_ACT_SCALE, W_SCALE = K_PLANES_W_SCALE;

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (fixed order), result in all four
__device__ __forceinline__ float quad_sum(float v) {
    float a, b;
    pope_xor16_pair(v, a, b);
    pope_xor32_pair(a + b, a, b);
    return a + b;
}

#ifdef RL_STAMPS  // dev: wall-clock stamps (10 ns ticks) of block 0, per kind of launch (K = 384 / 1536 / other), scripts/rowln_stamps.py
__device__ unsigned long long g_rl_dbg[3][64];
#define RL_STAMP(slot)                                                                                             \
    do {                                                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0 && rl_si + (slot) < 64)                                            \
            g_rl_dbg[g.K == 384 ? 0 : (g.K == 1536 ? 1 : 2)][rl_si + (slot)] = wall_clock64();                     \
    } while (0)
#else
#define RL_STAMP(slot) do {} while (0)
#endif

// LN_PLANES: LayerNorm output as activation planes (next GEMM's operand) or fp32 (final norm);
// RES_TABLE: the residual row is row % res_mod of a [res_mod, 384] table (patch embed: cls / conv bias + pos)
template <bool LN_PLANES, bool RES_TABLE>
__global__ __launch_bounds__(RTH, 2) void gemm_rowln16_kernel(const GemmParams g, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);
    float* stats = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + STATS_OFF);   // [2][4][128]
    // per-column constants of the epilogue, staged once per workgroup (they would otherwise hold 96 registers or put
    // loads between the epilogue's stores): gamma / scale, bias * gamma, ln_w, ln_b
    float* ctab = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + CTAB_OFF);     // [4][384]
    constexpr int NLD = 8;   // 16-byte pieces per thread and K-step: 2 A rows + 6 W rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int prow = tid >> 3, pc = tid & 7;
    const int nk = g.K / RK;  // >= 2 (launcher)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a_pl), 0, unsigned(g.M) * unsigned(g.lda) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_pl), 0, unsigned(RN) * unsigned(g.ldw) * 4u, 0x00020000);

    // tile stream: full rounds by XCD-remapped id, the partial last round by raw blockIdx (gemm_planes.hip)
    const int grid = gridDim.x, full_rounds = n_tiles / grid;
    const int remapped = xcd_remap(blockIdx.x, grid);
    const int tail_cand = full_rounds * grid + int(blockIdx.x);
    const int tail_tile = tail_cand < n_tiles ? tail_cand : n_tiles;
    auto tile_of = [&](int ord) -> int {
        const int in_tail = pope_uniform_select(ord == full_rounds, tail_tile, n_tiles);
        return pope_uniform_select(ord < full_rounds, ord * grid + remapped, in_tail);
    };
    const int first = tile_of(0);
    if (first >= n_tiles) return;
    if (tid < RN) {
        const float gm = g.gamma ? g.gamma[tid] : 1.0f;
        ctab[tid] = gm * (1.0f / (A_SCALE * W_SCALE));   // res + (v/scale + bias)*gamma = res + v*(gamma/scale) + bias*gamma
        ctab[RN + tid] = (g.bias ? g.bias[tid] : 0.f) * gm;
        ctab[2 * RN + tid] = g.ln_w[tid];
        ctab[3 * RN + tid] = g.ln_b[tid];
    }
    // one VGPR offset per operand: the row block (64 i rows), the tile and the K-step travel in the scalar offset
    const unsigned va = unsigned(prow) * unsigned(g.lda) * 4u + pc * 16u, vw = unsigned(prow) * unsigned(g.ldw) * 4u + pc * 16u;
    const unsigned a64 = 64u * unsigned(g.lda) * 4u, w64 = 64u * unsigned(g.ldw) * 4u;
    u32x4 r0[NLD];  // A rows, then W rows: the K-step after the one in the other LDS stage
    int ld_ord = 0, ord = 0;
    int ld_tile = first, ld_kt = 0;
    auto load_next = [&]() {
        const int lt = ld_tile < n_tiles ? ld_tile : n_tiles - 1;   // past the end: re-load, never consumed
        const unsigned sa = unsigned(lt) * unsigned(RM) * unsigned(g.lda) * 4u + unsigned(ld_kt) * 128u, sw = unsigned(ld_kt) * 128u;
        // A rows are read exactly once per launch (a tile spans all 384 columns): sc0 + nt keeps the 150 - 600 MB stream
        // from displacing x / xn, which the next kernels re-read (+0.8 % on the step; the same hint on the residual
        // loads, the x stores or the xn stores costs 1 - 10 %: measured, left at the default policy)
#pragma unroll
        for (int i = 0; i < 2; ++i) r0[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, va, sa + i * a64, 3);
#pragma unroll
        for (int i = 0; i < 6; ++i) r0[2 + i] = __builtin_amdgcn_raw_buffer_load_b128(rw, vw, sw + i * w64, 0);
        const int wrap = ++ld_kt == nk;
        ld_kt = pope_uniform_select(wrap, 0, ld_kt);
        ld_ord += wrap;
        ld_tile = tile_of(ld_ord);
    };
    // LDS image of a K-step: row r of an operand at r * 128 B, its 16-byte chunk c at chunk c ^ (r & 7)
    const int wr_off = prow * ROWB + 8 * (pc ^ (prow & 7));   // rows prow + 64 i: same r & 7
    auto write_stage = [&](int s) {
        _Float16* S = lds + s * STAGE_H + wr_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(S + 64 * i * ROWB) = r0[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(S + (RM + 64 * i) * ROWB) = r0[2 + i];
    };
    // fragment t: rows 16 t + l15 of this wave's rows, logical chunk plane * 4 + q4
    const int sw_hi = 8 * (q4 ^ (l15 & 7)), sw_lo = 8 * ((4 + q4) ^ (l15 & 7));
    const int a_row = (wm * 64 + l15) * ROWB, w_row = (RM + wn * 96 + l15) * ROWB;
    f32x4 acc[4][6];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, unsigned(g.M) * unsigned(RN) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rln = __builtin_amdgcn_make_buffer_rsrc(LN_PLANES ? g.ln_planes : static_cast<void*>(g.ln_f32), 0,
                                                                         unsigned(g.M) * unsigned(RN) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.res), 0, unsigned(RES_TABLE ? g.res_mod : g.M) * unsigned(g.ldres) * 4u, 0x00020000);

    // Residual prefetch.  The epilogue's 590 KB per tile (residual in, x and xn out) hit HBM from all 256 CUs at once while
    // the K loops leave it idle; the residual third of that burst is pulled forward: during the K loop every lane touches
    // three of the tile's 1 536 residual lines (one dword each, loaded straight into an LDS landing pad that nobody reads —
    // no register, no wait), so the epilogue's residual loads find their lines in L2 / MALL.
    lds_void_ptr pf_pad = (lds_void_ptr)(reinterpret_cast<char*>(smem) + PF_OFF + (tid >> 6) * 256);
    const int pf_step = nk / 3;
    auto prefetch_res = [&](int tile_, int j) {
        const unsigned L = unsigned(tid) + 512u * unsigned(j), row = L / 12u, piece = L - 12u * row;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rres, pf_pad, 4, (unsigned(tile_) * RM + row) * unsigned(RN) * 4u + piece * 128u, 0, 0, 0);
    };

    int rl_si = 0;   // stamp index (dev builds)
    auto epilogue = [&](int tile, int free_stage) {
        const int m0 = tile * RM;
        // per-wave transposition buffer in the LDS stage the K loop has just finished with (the other stage already holds
        // the next tile's first K-step): 16 rows x 96 columns of this wave at a time, 400-byte pitch (conflict-free
        // 16-byte pieces from the accumulator layout), read back as whole 128-byte lines: 8 lanes per line
        float* wreg = reinterpret_cast<float*>(lds + free_stage * STAGE_H) + (threadIdx.x >> 6) * 1600;
        int elane = threadIdx.x & 63;
        asm volatile("" : "+v"(elane));
        const int epiece = elane & 7;
        RL_STAMP(1);
        // The lane coordinates are re-derived behind an opaque fence: every address below would otherwise be hoisted
        // out of the tile loop as a loop invariant (~40 registers held across the K-steps: spills in the mainloop).
        int l15 = threadIdx.x & 15, q4 = (threadIdx.x >> 4) & 3;
        asm volatile("" : "+v"(l15), "+v"(q4));
        // pin everything below behind the tile-end branch
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
        // ---- 1. x = res + gamma * (acc / scale + bias), in place --------------------------------------------------
        const int col0 = wn * 96 + 4 * q4;   // this lane's columns: col0 + 16 ni .. + 3
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) {
            const f32x4 gam = *reinterpret_cast<const f32x4*>(ctab + col0 + 16 * ni);
            const f32x4 bia = *reinterpret_cast<const f32x4*>(ctab + RN + col0 + 16 * ni);
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) {
                const unsigned row = unsigned(m0 + wm * 64 + mi * 16 + l15);
                const unsigned rr = RES_TABLE ? row % unsigned(g.res_mod) : row;   // rows >= M: out of range -> 0
                const f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rres, rr * unsigned(g.ldres) * 4u + unsigned(col0) * 4u, ni * 64, 0));
                acc[mi][ni] = r + acc[mi][ni] * gam + bia;
            }
            __builtin_amdgcn_sched_barrier(0);   // one column block's residual rows in flight (2, 3 or 6: no difference —
                                                 // the phase runs at the memory system's speed, all CUs at once)
        }
        __builtin_amdgcn_sched_barrier(0);
        RL_STAMP(2);
        // ---- 2. row means: lane -> quad of lanes -> the four column waves (LDS) ------------------------------------
        const int rl = wm * 64 + l15;   // row within the tile, + 16 mi
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            float s = 0.f;
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) s += (acc[mi][ni][0] + acc[mi][ni][1]) + (acc[mi][ni][2] + acc[mi][ni][3]);
            s = quad_sum(s);
            if (q4 == 0) stats[wn * RM + rl + 16 * mi] = s;
        }
        __syncthreads();
        float mean[NMI], rstd[NMI];
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            const float* p = stats + rl + 16 * mi;
            mean[mi] = ((p[0] + p[RM]) + (p[2 * RM] + p[3 * RM])) * (1.0f / float(RN));
        }
        __builtin_amdgcn_sched_barrier(0);
        RL_STAMP(3);
        // ---- 3. x to memory (the residual stream), then centre in place and take the second moment -----------------
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) *reinterpret_cast<f32x4*>(&wreg[l15 * 100 + ni * 16 + 4 * q4]) = acc[mi][ni];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 6; ++t) {   // 48 lines of 128 bytes: 16 rows x 3
                const int L = t * 8 + (elane >> 3), row = (L * 43) >> 7, ln = L - 3 * row;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&wreg[row * 100 + ln * 32 + epiece * 4]);
                const unsigned off = unsigned(m0 + wm * 64 + mi * 16 + row) * unsigned(RN) * 4u + unsigned(wn * 96 + ln * 32 + epiece * 4) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rx, off, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            float qs = 0.f;
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) {
                const f32x4 d = acc[mi][ni] - mean[mi];
                acc[mi][ni] = d;
                qs += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            qs = quad_sum(qs);
            if (q4 == 0) stats[4 * RM + wn * RM + rl + 16 * mi] = qs;
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            const float* p = stats + 4 * RM + rl + 16 * mi;
            const float var = ((p[0] + p[RM]) + (p[2 * RM] + p[3 * RM])) * (1.0f / float(RN));
            rstd[mi] = 1.0f / sqrtf(var + g.ln_eps);
        }
        __builtin_amdgcn_sched_barrier(0);
        RL_STAMP(4);
        float finite_probe = 0.f;   // a non-finite row (poisoned x) has a non-finite mean or rstd
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) finite_probe += __builtin_fabsf(mean[mi]) + rstd[mi];
        // ---- 4. xn = (x - mean) * rstd * w + b -> planes (or fp32) ---------------------------------------------------
        f32x2 amax = {0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) {
                const f32x4 lw = *reinterpret_cast<const f32x4*>(ctab + 2 * RN + col0 + 16 * ni);
                const f32x4 lb = *reinterpret_cast<const f32x4*>(ctab + 3 * RN + col0 + 16 * ni);
                const f32x4 y = acc[mi][ni] * rstd[mi] * lw + lb;
                if constexpr (LN_PLANES) {
                    const f32x4 ys = y * A_SCALE;
                    pope_amax4x2(amax, ys);
                    f16x4 hi, lo;
                    pope_split4(ys, hi, lo);
                    const int c = ni * 16 + 4 * q4;   // column within the wave's 96 = three planes chunks of 128 bytes
                    _Float16* hp = reinterpret_cast<_Float16*>(wreg) + l15 * 200 + (c >> 5) * 64 + (c & 31);
                    *reinterpret_cast<f16x4*>(hp) = hi;
                    *reinterpret_cast<f16x4*>(hp + 32) = lo;
                } else {
                    *reinterpret_cast<f32x4*>(&wreg[l15 * 100 + ni * 16 + 4 * q4]) = y;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int L = t * 8 + (elane >> 3), row = (L * 43) >> 7, ln = L - 3 * row;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&wreg[row * 100 + ln * 32 + epiece * 4]);
                // planes rows and fp32 rows have the same pitch, and the wave's 96 columns are 384 bytes of either
                const unsigned off = unsigned(m0 + wm * 64 + mi * 16 + row) * unsigned(RN) * 4u + unsigned(wn * 96 + ln * 32 + epiece * 4) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rln, off, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (LN_PLANES)   // a non-finite row (poisoned x) has a non-finite mean or rstd; fmax ignores NaN
            pope_range_flag(g.range_flag, POPE_RANGE_LAYERNORM,
                            !(__builtin_fmaxf(amax[0], amax[1]) < POPE_F16_OVERFLOW) ||
                                !(finite_probe < INFINITY));
        RL_STAMP(5);
    };

    // prologue: item 0 -> LDS stage 0; item 1 in flight
    load_next();
    write_stage(0);
    load_next();
    __syncthreads();
    zero_acc();
    int tile = first, kt = 0;
    RL_STAMP(0);

    auto item = [&](int s) {
        const _Float16* S = lds + (s & 1) * STAGE_H;
        // ni-major: the A fragments (hi, lo: 8 x 4 registers) stay for the K-step, the W fragments stream through two at
        // a time (lo, hi of column block ni), each feeding 12 MFMAs — 48 fragment registers instead of 80 (the kernel
        // sits at the 256-register line: 96 accumulators + 32 staging).  Per accumulator the order of the partial
        // products is that of gemm_planes.hip (lo.hi, hi.lo, hi.hi): bit-identical sums.
        // Program order is pinned with scheduling fences: the scheduler otherwise hoists all 20 fragment reads to the
        // top of the K-step (80 live registers) and spills the in-flight staging registers to scratch.
        f16x8 ah[NMI], al[NMI], wl[2], wh[2];
#pragma unroll
        for (int t = 0; t < NMI; ++t) {
            ah[t] = *reinterpret_cast<const f16x8*>(S + a_row + t * 16 * ROWB + sw_hi);
            al[t] = *reinterpret_cast<const f16x8*>(S + a_row + t * 16 * ROWB + sw_lo);
        }
        wl[0] = *reinterpret_cast<const f16x8*>(S + w_row + sw_lo);
        wh[0] = *reinterpret_cast<const f16x8*>(S + w_row + sw_hi);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) {
            const int cur = ni & 1, nxt = cur ^ 1;
            if (ni + 1 < 6) {
                wl[nxt] = *reinterpret_cast<const f16x8*>(S + w_row + (ni + 1) * 16 * ROWB + sw_lo);
                wh[nxt] = *reinterpret_cast<const f16x8*>(S + w_row + (ni + 1) * 16 * ROWB + sw_hi);
            }
            // item s+1 goes to LDS late in the K-step and item s+2 is requested right after it: the staging registers
            // are loaded for almost a whole K-step — except across a tile seam: the epilogue needs those registers (with
            // them live it spills, and a spill reload drains vmcnt, i.e. waits for every load in flight), so the seam
            // load is issued after the epilogue and still has four column groups of MFMAs to arrive
            if (ni == 4) write_stage((s + 1) & 1);
            if (ni == 5 && kt + 1 != nk) load_next();
            if constexpr (!RES_TABLE)
                if (ni == 2 && g.rl_prefetch && pf_step > 0 && kt % pf_step == 0 && kt / pf_step < 3) prefetch_res(tile, kt / pf_step);
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) acc[mi][ni] = mfma16(wl[cur], ah[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) acc[mi][ni] = mfma16(wh[cur], al[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) acc[mi][ni] = mfma16(wh[cur], ah[mi], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // stage (s+1)&1 is published, stage s&1 is free
        if (++kt == nk) {
            epilogue(tile, s & 1);
            __syncthreads();   // the next K-step stores into the stage the slower waves may still be transposing through
            load_next();   // the K-step after the one already in LDS
            zero_acc();
            kt = 0;
            tile = tile_of(++ord);
            rl_si += 8;
            RL_STAMP(0);
        }
    };
    for (int s = 0; tile < n_tiles; ++s) item(s);
}

template <bool LN_PLANES, bool RES_TABLE>
int launch_rowln(const GemmParams& g, hipStream_t stream) {
    static pope_dev_mask lds_ok{0};
    if (!pope_opt_in_lds(gemm_rowln16_kernel<LN_PLANES, RES_TABLE>, RL_LDS_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = (g.M + RM - 1) / RM, cus = pope_cu_count();
    hipLaunchKernelGGL((gemm_rowln16_kernel<LN_PLANES, RES_TABLE>), dim3(tiles < cus ? tiles : cus), dim3(RTH), RL_LDS_BYTES, stream, g, tiles);
    return pope_check_launch();
}

}  // namespace

#ifdef RL_STAMPS
extern "C" int pope_lab_rowln_stamps(unsigned long long* host192) {
    return hipMemcpyFromSymbol(host192, HIP_SYMBOL(g_rl_dbg), sizeof(unsigned long long) * 192) == hipSuccess ? 0 : -1;
}
#endif

bool pope_gemm_rowln_supported(const GemmParams& g) {
    return g.N == RN && g.ldc == RN && g.K >= 2 * RK && (g.K % RK) == 0 && g.lda == g.K && g.ldw == g.K && g.ldres == RN &&
           size_t(g.M + RM) * g.lda * 4 < (size_t(1) << 32) && size_t(g.M + RM) * RN * 4 < (size_t(1) << 32) - 512;
}

// x = res + gamma * (A.W^T + bias) -> g.C (fp32, may alias res), LayerNorm(x; ln_w, ln_b, ln_eps) -> g.ln_planes or g.ln_f32
int pope_launch_gemm_rowln(const GemmParams& g_in, hipStream_t stream) {
    GemmParams g = g_in;
    g.rl_prefetch = 1;   // touch the tile's residual lines during its K loop (finding 21: proj -3 %)
    if (!g.a_pl || !g.w_pl || !g.C || !g.res || !g.ln_w || !g.ln_b || (!g.ln_planes) == (!g.ln_f32) || g.M <= 0) return POPE_ERR_ARG;
    if (!pope_gemm_rowln_supported(g)) return POPE_ERR_ARG;
    if (g.res_mod < 0 || (g.res_mod == 0 && !g.gamma)) return POPE_ERR_ARG;
    if (g.res_mod > 0) return g.ln_planes ? launch_rowln<true, true>(g, stream) : launch_rowln<false, true>(g, stream);
    return g.ln_planes ? launch_rowln<true, false>(g, stream) : launch_rowln<false, false>(g, stream);
}
