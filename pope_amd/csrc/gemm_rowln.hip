// Residual GEMM with the FOLLOWING LayerNorm fused into its epilogue (north star: "fused LayerNorm + projection"):
//     x  <- res + gamma * (A . W^T + bias)                       block.py:105-106 (proj / fc2); patch embed (res = pos table)
//     xn <- LayerNorm(x) * ln_w + ln_b   (eps inside the sqrt)   block.py:56,68 (norm2 / next block's norm1), final norm
// for N = 384 (ViT-S/14).  x is written as fp32 (the residual stream), xn as f16x3 activation planes for the next
// GEMM (or as fp32 for the final norm): the stand-alone LayerNorm launches between the GEMMs — 100 per step, each a
// read of x and a write of the planes at the HBM copy ceiling — disappear; x is normalised while its tile is still
// in the producing workgroup's registers.
//
// A LayerNorm row needs all 384 columns, so the tile is a FULL ROW BLOCK: RM rows x 384 columns, one workgroup of 8 waves per
// CU, persistent over the tile list.  Two geometries (RlGeo), chosen per launch by rounds x rows: 192 rows (wave (wm, wn) owns
// 96 rows x 96 columns = 6 x 6 accumulator blocks of v_mfma_f32_16x16x32_f16; 511 tiles = 2 rounds for a 64-image chunk) and
// 128 rows (64 x 96 per wave; 766 tiles = 3 rounds).  Operands are planes; a K-step's 128-byte row pieces go memory -> LDS
// DIRECTLY (buffer_load ... lds, round 4: no staging registers, no ds_write; the VGPR -> LDS path was what this kernel's
// K-step waited on, DESIGN.md finding 20), into unpadded rows whose 16-byte chunk index is XOR-ed with row & 7 on the source
// side: conflict-free 16-row fragment reads; two stages of 64 / 72 KB, ONE barrier per K-step (LDS-scope fence only: a full
// __syncthreads() would also wait for the epilogue's global stores).  The K-steps of consecutive tiles form one stream: the
// first K-step of the next tile lands during the epilogue.  With no staging registers the kernel needs 150 - 220 VGPRs instead
// of 256 with zero slack: every per-lane loop invariant is re-derived per K-step behind an opaque fence so that the allocator
// has nothing to spill across the epilogue (a spill reload in the K loop waits for vmcnt(0), i.e. for the loads just issued).
// Measured (same box, profiles/r04/rowln_dma_ab.txt): FC2+LN 0.400 -> 0.364 ms, proj+LN 0.159 -> 0.150, patch embed 0.288 -> 0.268.
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

constexpr int RN = 384, RK = 32;
constexpr int ROWB = 64;                          // halves per LDS row: 128 B, chunk-swizzled
// Tile geometry: RM rows x 384 columns, NWM x 4 waves (a wave owns RM / NWM rows x 96 columns), one workgroup per CU
template <int RM_, int NWM_> struct RlGeo {
    static constexpr int RM = RM_, NWM = NWM_, NW = 4 * NWM, RTH = 64 * NW;
    static constexpr int NMI = RM / NWM / 16;          // 16-row accumulator blocks per wave
    static constexpr int STAGE_H = (RM + RN) * ROWB;   // halves per stage (A rows, then W rows): 64 KB / 72 KB
    static constexpr size_t STATS_OFF = size_t(2) * STAGE_H * sizeof(_Float16);
    static constexpr size_t CTAB_OFF = STATS_OFF + size_t(2) * 4 * RM * sizeof(float);   // after the two [4][RM] row-sum tables
    static constexpr size_t PF_OFF = CTAB_OFF + size_t(4) * RN * sizeof(float);          // + per-column constants
    static constexpr size_t LDS_BYTES = PF_OFF + size_t(NW) * 256;                       // + landing pads of the residual prefetch
    // rows of a 16-row accumulator block that go through a wave's transposition buffer at a time: the NW buffers (100 floats
    // per row) live in the stage the K loop has just left
    static constexpr int EROWS = size_t(NW) * 16 * 100 * sizeof(float) <= size_t(STAGE_H) * sizeof(_Float16) ? 16 : 8;
    static constexpr int AROWS = RM / NW, WROWS = RN / NW;   // rows a wave stages per K-step
    static constexpr int NPF = (RM * 12 + RTH - 1) / RTH;    // residual-prefetch rounds per tile: one 128-byte line per lane each
    static_assert(size_t(NW) * EROWS * 100 * sizeof(float) <= size_t(STAGE_H) * sizeof(_Float16), "transposition buffers must fit a stage");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");
    static_assert(WROWS % 8 == 0 && AROWS % 8 == 0 && RM == NWM * NMI * 16, "eight rows per staging instruction");
};
typedef __attribute__((address_space(3))) void* lds_void_ptr;
constexpr float A_SCALE = K_PLANES_ACT_SCALE, W_SCALE = K_PLANES_W_SCALE;

// lane id from the execution mask counters, seeded with an opaque zero: no input register (threadIdx.x held across the tile loop
// gets spilled, and its reload in every K-step waits for vmcnt(0)), and not loop-invariant either (hoisted, the result is spilled too)
__device__ __forceinline__ int pope_lane_id() {
    unsigned z = 0;
    asm volatile("" : "+s"(z));
    return int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)));
}
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
        if (LN && blockIdx.x == 0 && threadIdx.x == 0 && rl_si + (slot) < 64)   /* the LayerNorm modes only */     \
            g_rl_dbg[g.K == 384 ? 0 : (g.K == 1536 ? 1 : 2)][rl_si + (slot)] = wall_clock64();                     \
    } while (0)
#else
#define RL_STAMP(slot) do {} while (0)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a full workgroup-scope fence: the compiler puts
// s_waitcnt vmcnt(0) in front of it, i.e. every global store of the epilogue in flight must be acknowledged before the wave
// may even arrive (2 - 3 us per tile seam, and in the middle of the epilogue between the x stores and the second moment).
// The LDS-direct loads are waited for explicitly where a stage is published.
__device__ __forceinline__ void rl_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// MODE: what leaves the tile.  RL_LN_PLANES / RL_LN_F32: x (fp32) and LayerNorm(x) as activation planes / fp32 — ONE column
// tile (N = 384); RL_BIAS_PLANES / RL_GELU_PLANES (round 4): A.W^T + bias (+ exact-erf GELU) as activation planes, any number
// of 384-column tiles (QKV: 3, FC1: 4 — the ViT-S/14 widths are multiples of 384, so the 192 x 384 stream has no partial
// column tile, where 256 x 256 tiles compute 1 280 columns for QKV's 1 152), in gemm_plain.hip's epilogue arithmetic.
enum { RL_LN_PLANES = 0, RL_LN_F32 = 1, RL_BIAS_PLANES = 2, RL_GELU_PLANES = 3 };

// exact-erf GELU on a pair: the arithmetic of gemm_planes.hip:gelu_erf_pair, instruction for instruction (bit-identical)
__device__ __forceinline__ f32x2 rl_gelu_pair(f32x2 x) {
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

// RES_TABLE (LayerNorm modes): the residual row is row % res_mod of a [res_mod, 384] table (patch embed: cls / conv bias + pos)
template <class G, int MODE, bool RES_TABLE>
__global__ __launch_bounds__(G::RTH) void gemm_rowln16_kernel(const GemmParams g, int n_tiles, int ncol) {
    constexpr bool LN = MODE == RL_LN_PLANES || MODE == RL_LN_F32, LN_PLANES = MODE == RL_LN_PLANES;
    constexpr int RM = G::RM, NMI = G::NMI, RTH = G::RTH, STAGE_H = G::STAGE_H, EROWS = G::EROWS, AROWS = G::AROWS, WROWS = G::WROWS;
    constexpr int WMR = 16 * NMI;                      // rows of a wave
    constexpr int NIA = AROWS / 8, NIW = WROWS / 8;    // staging instructions per wave and K-step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);
    float* stats = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + G::STATS_OFF);   // [2][4][RM]
    // per-column constants of the epilogue, staged once per workgroup (they would otherwise hold 96 registers or put
    // loads between the epilogue's stores): gamma / scale, bias * gamma, ln_w, ln_b
    float* ctab = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + G::CTAB_OFF);     // [4][384]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = g.K / RK;  // >= 2 (launcher)
    const unsigned lda4 = unsigned(g.lda) * 4u, ldw4 = unsigned(g.ldw) * 4u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a_pl), 0, unsigned(g.M) * lda4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_pl), 0, unsigned(g.N) * ldw4, 0x00020000);
    // tile id -> (row tile, column tile): the column tiles of a row tile are neighbours in the stream (the A rows leave HBM once)
    auto row_tile = [&](int t) -> int { return LN ? t : t / ncol; };
    auto col_tile = [&](int t) -> int { return LN ? 0 : t - (t / ncol) * ncol; };

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
    if (LN && tid < RN) {
        const float gm = g.gamma ? g.gamma[tid] : 1.0f;
        ctab[tid] = gm * (1.0f / (A_SCALE * W_SCALE));   // res + (v/scale + bias)*gamma = res + v*(gamma/scale) + bias*gamma
        ctab[RN + tid] = (g.bias ? g.bias[tid] : 0.f) * gm;
        ctab[2 * RN + tid] = g.ln_w[tid];
        ctab[3 * RN + tid] = g.ln_b[tid];
    }

    // ---- staging: memory -> LDS directly (buffer_load ... lds: no staging registers, no ds_write; gemm_plain.hip).  A
    // wave-instruction moves 8 rows x 128 B; lane (r8 = lane >> 3, position lane & 7) fetches the 16-byte piece that belongs at
    // its position of the swizzled row (piece c of row r sits at c ^ (r & 7); the rows of an instruction start at a multiple
    // of 8).  The tile's row offset travels in the VGPR offset — the descriptor's range check then returns zeros for the rows
    // past M of the last tile (a scalar offset is not range-checked) — W's row blocks and the K-step in the scalar offset.
    // A rows are read exactly once per launch (a tile spans all 384 columns): sc0 + nt keeps the 150 - 600 MB stream from
    // displacing x / xn, which the next kernels re-read.
    // Per-lane loop invariants (staging offsets, fragment addresses) are RE-DERIVED from the lane id behind an opaque fence in
    // every K-step (a dozen VALU instructions): held in registers across the tile loop they are the allocator's first spill
    // candidates under the epilogue's pressure, and a spill reload inside the K loop waits for vmcnt(0) — for the loads just issued.
    const unsigned a8 = 8u * lda4, w8 = 8u * ldw4;
    auto dma = [&](int stage, int t, int kt_) {
        int ln_ = pope_lane_id();
        asm volatile("" : "+v"(ln_));
        const int r8 = ln_ >> 3, piece = (ln_ & 7) ^ r8;
        const int lt = t < n_tiles ? t : n_tiles - 1;   // past the end: re-load, never consumed
        _Float16* S_ = lds + stage * STAGE_H;
        const unsigned vt = unsigned(row_tile(lt) * RM + AROWS * wave + r8) * lda4 + unsigned(piece) * 16u;
        const unsigned ko = unsigned(kt_) * 128u + unsigned(col_tile(lt)) * unsigned(RN) * ldw4;   // (zero column offset for the A loads below: LN)
        const unsigned vw0 = unsigned(WROWS * wave + r8) * ldw4 + unsigned(piece) * 16u;
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(S_ + (AROWS * wave + 8 * i) * ROWB), 16, vt + i * a8, unsigned(kt_) * 128u, 0, LN ? 3 : 0);   // (planes modes: the column siblings share the rows through L2; nt there: QKV +4.5 % slower)
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_ptr)(S_ + (RM + WROWS * wave + 8 * i) * ROWB), 16, vw0, ko + i * w8, 0, 0);
    };
    f32x4 acc[NMI][6];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // (LN modes; in the planes modes these descriptors are never used)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, unsigned(g.M) * unsigned(RN) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rln = __builtin_amdgcn_make_buffer_rsrc(LN_PLANES ? g.ln_planes : static_cast<void*>(g.ln_f32), 0,
                                                                         unsigned(g.M) * unsigned(RN) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.res), 0, unsigned(RES_TABLE ? g.res_mod : g.M) * unsigned(g.ldres) * 4u, 0x00020000);

    // Residual prefetch.  The epilogue's bytes (residual in, x and xn out: 590 KB per 128 rows) hit HBM from all 256 CUs at
    // once while the K loops leave it idle; the residual third of that burst is pulled forward: in the LAST K-steps of the tile
    // (one round per K-step; spread over the whole K loop the first lines are 100+ us old by the epilogue: FC2 +1.7 % slower)
    // every lane touches one of the tile's 12 RM residual lines per round (one dword, loaded straight into an LDS landing pad
    // that nobody reads — no register, no wait), so the epilogue's residual loads find their lines in L2 / MALL.
    lds_void_ptr pf_pad = (lds_void_ptr)(reinterpret_cast<char*>(smem) + G::PF_OFF + wave * 256);
    auto prefetch_res = [&](int tile_, int j) {
        int t_ = wave * 64 + pope_lane_id();
        asm volatile("" : "+v"(t_));
        const unsigned L = unsigned(t_) + unsigned(RTH) * unsigned(j), row = L / 12u, pc = L - 12u * row;
        // (lines past the tile's last — the final round of a geometry whose line count is no multiple of RTH — fall outside the descriptor)
        const unsigned off = RM * 12 % RTH == 0 || L < unsigned(RM * 12) ? (unsigned(tile_) * RM + row) * unsigned(RN) * 4u + pc * 128u : 0xFFFFFF00u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rres, pf_pad, 4, off, 0, 0, 0);
    };

    int rl_si = 0;   // stamp index (dev builds)
    [[maybe_unused]] auto epilogue = [&](int tile, int free_stage) {
        const int m0 = tile * RM;
        // per-wave transposition buffer in the LDS stage the K loop has just finished with (the other stage is receiving
        // the next tile's first K-step): EROWS rows x 96 columns of this wave at a time, 400-byte pitch (conflict-free
        // 16-byte pieces from the accumulator layout), read back as whole 128-byte lines: 8 lanes per line
        float* wreg = reinterpret_cast<float*>(lds + free_stage * STAGE_H) + wave * (EROWS * 100);
        int elane = pope_lane_id();
        asm volatile("" : "+v"(elane));
        RL_STAMP(1);
        // The lane coordinates are re-derived behind an opaque fence: every address below would otherwise be hoisted
        // out of the tile loop as a loop invariant (~40 registers held across the K-steps: spills in the mainloop).
        int l15 = elane & 15, q4 = elane >> 4;
        asm volatile("" : "+v"(l15), "+v"(q4));
        const int er = l15 & (EROWS - 1);        // this lane's row of the transposition buffer
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
                const unsigned row = unsigned(m0 + wm * WMR + mi * 16 + l15);
                const unsigned rr = RES_TABLE ? row % unsigned(g.res_mod) : row;   // rows >= M: out of range -> 0
                const f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rres, rr * unsigned(g.ldres) * 4u + unsigned(col0) * 4u, ni * 64, 0));
                acc[mi][ni] = r + acc[mi][ni] * gam + bia;
            }
            // x is formed HERE: without the pin the middle end sinks the arithmetic of the later row blocks into the branches
            // of phase 2 and keeps the loaded residual rows alive until then (spilled, each behind a vmcnt(0))
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) asm volatile("" : "+v"(acc[mi][ni]));
            __builtin_amdgcn_sched_barrier(0);   // one column block's residual rows in flight (2, 3 or 6: no difference —
                                                 // the phase runs at the memory system's speed, all CUs at once)
        }
        __builtin_amdgcn_sched_barrier(0);
        RL_STAMP(2);
        // ---- 2. row means: lane -> quad of lanes -> the four column waves (LDS) ------------------------------------
        const int rl = wm * WMR + l15;   // row within the tile, + 16 mi
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            float s = 0.f;
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) s += (acc[mi][ni][0] + acc[mi][ni][1]) + (acc[mi][ni][2] + acc[mi][ni][3]);
            s = quad_sum(s);
            if (q4 == 0) stats[wn * RM + rl + 16 * mi] = s;
        }
        rl_barrier();
        float mean[NMI], rstd[NMI];
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            const float* p = stats + rl + 16 * mi;
            mean[mi] = ((p[0] + p[RM]) + (p[2 * RM] + p[3 * RM])) * (1.0f / float(RN));
        }
        __builtin_amdgcn_sched_barrier(0);
        RL_STAMP(3);
        // a pass of the transposition buffer: EROWS rows x 3 lines of 128 bytes, 8 lines per wave-instruction
        auto store_lines = [&](const __amdgpu_buffer_rsrc_t& rdst, int row_base) {
            // (coordinates re-derived per pass: hoisted, the x pass and the xn pass share 64-bit address pairs that the
            // allocator spills and reloads behind vmcnt(0), i.e. behind every store in flight)
            int el = elane;
            asm volatile("" : "+v"(el));
            const int epiece = el & 7;
#pragma unroll
            for (int t = 0; t < (EROWS * 3) / 8; ++t) {
                const int L = t * 8 + (el >> 3), row = (L * 43) >> 7, ln = L - 3 * row;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&wreg[row * 100 + ln * 32 + epiece * 4]);
                // planes rows and fp32 rows have the same pitch, and the wave's 96 columns are 384 bytes of either
                const unsigned off = unsigned(row_base + row) * unsigned(RN) * 4u + unsigned(wn * 96 + ln * 32 + epiece * 4) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rdst, off, 0, 0);
            }
        };
        // ---- 3. x to memory (the residual stream), then centre in place and take the second moment -----------------
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
#pragma unroll
            for (int h = 0; h < 16 / EROWS; ++h) {
                if (EROWS == 16 || (l15 >> 3) == h) {
#pragma unroll
                    for (int ni = 0; ni < 6; ++ni) *reinterpret_cast<f32x4*>(&wreg[er * 100 + ni * 16 + 4 * q4]) = acc[mi][ni];
                }
                __builtin_amdgcn_wave_barrier();
                store_lines(rx, m0 + wm * WMR + mi * 16 + h * EROWS);
                __builtin_amdgcn_wave_barrier();
            }
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
        rl_barrier();
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
            for (int h = 0; h < 16 / EROWS; ++h) {
                if (EROWS == 16 || (l15 >> 3) == h) {   // (8-row passes: a lane's quads go out in ONE of the two)
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
                            _Float16* hp = reinterpret_cast<_Float16*>(wreg) + er * 200 + (c >> 5) * 64 + (c & 31);
                            *reinterpret_cast<f16x4*>(hp) = hi;
                            *reinterpret_cast<f16x4*>(hp + 32) = lo;
                        } else {
                            *reinterpret_cast<f32x4*>(&wreg[er * 100 + ni * 16 + 4 * q4]) = y;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                store_lines(rln, m0 + wm * WMR + mi * 16 + h * EROWS);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (LN_PLANES)   // a non-finite row (poisoned x) has a non-finite mean or rstd; fmax ignores NaN
            pope_range_flag(g.range_flag, POPE_RANGE_LAYERNORM,
                            !(__builtin_fmaxf(amax[0], amax[1]) < POPE_F16_OVERFLOW) ||
                                !(finite_probe < INFINITY));
        RL_STAMP(5);
    };

    // ---- planes epilogue (RL_BIAS_PLANES / RL_GELU_PLANES): y = acc / scale + bias [-> GELU] -> activation planes, through the
    // same per-wave transposition buffer and whole-line stores; the arithmetic is gemm_plain.hip's (bit-identical outputs)
    const __amdgpu_buffer_rsrc_t rcp = __builtin_amdgcn_make_buffer_rsrc(LN ? static_cast<void*>(g.C) : g.c_pl, 0,
                                                                         unsigned(g.M) * unsigned(g.ldc) * 4u, 0x00020000);
    [[maybe_unused]] auto epilogue_planes = [&](int tile, int free_stage) {
        const int m0 = row_tile(tile) * RM, n0 = col_tile(tile) * RN;
        float* wreg = reinterpret_cast<float*>(lds + free_stage * STAGE_H) + wave * (EROWS * 100);
        int elane = pope_lane_id();
        asm volatile("" : "+v"(elane));
        int l15 = elane & 15, q4 = elane >> 4;
        asm volatile("" : "+v"(l15), "+v"(q4));
        const int er = l15 & (EROWS - 1);
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
        const int col0 = n0 + wn * 96 + 4 * q4;   // this lane's columns: col0 + 16 ni .. + 3
        constexpr float inv = 1.0f / (A_SCALE * W_SCALE);
        f32x4 bias[6];   // loaded before the first store of the epilogue (a load between stores waits for every store in flight)
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) bias[ni] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + col0 + 16 * ni) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x2 amax = {0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
#pragma unroll
            for (int h = 0; h < 16 / EROWS; ++h) {
                if (EROWS == 16 || (l15 >> 3) == h) {
#pragma unroll
                    for (int ni = 0; ni < 6; ++ni) {
                        f32x4 v = acc[mi][ni] * inv + bias[ni];
                        if constexpr (MODE == RL_GELU_PLANES) {
                            const f32x2 g01 = rl_gelu_pair(f32x2{v[0], v[1]}), g23 = rl_gelu_pair(f32x2{v[2], v[3]});
                            v = f32x4{g01[0], g01[1], g23[0], g23[1]};
                        }
                        pope_amax4x2(amax, v);
                        f16x4 hi, lo;
                        pope_split4(v * A_SCALE, hi, lo);
                        const int c = ni * 16 + 4 * q4;   // column within the wave's 96 = three planes chunks of 128 bytes
                        _Float16* hp = reinterpret_cast<_Float16*>(wreg) + er * 200 + (c >> 5) * 64 + (c & 31);
                        *reinterpret_cast<f16x4*>(hp) = hi;
                        *reinterpret_cast<f16x4*>(hp + 32) = lo;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                {   // EROWS rows x 3 lines of 128 bytes, 8 lines per wave-instruction (write-once output: non-temporal)
                    int el = elane;
                    asm volatile("" : "+v"(el));
                    const int epiece = el & 7;
                    const int row_base = m0 + wm * WMR + mi * 16 + h * EROWS;
#pragma unroll
                    for (int t = 0; t < (EROWS * 3) / 8; ++t) {
                        const int L = t * 8 + (el >> 3), row = (L * 43) >> 7, ln = L - 3 * row;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&wreg[row * 100 + ln * 32 + epiece * 4]);
                        const unsigned off = unsigned(row_base + row) * unsigned(g.ldc) * 4u + unsigned(n0 + wn * 96 + ln * 32 + epiece * 4) * 4u;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rcp, off, 0, 2);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) * A_SCALE < POPE_F16_OVERFLOW));
    };

    // prologue: the first K-step of the first tile -> stage 0
    dma(0, first, 0);
    zero_acc();
    int tile = first, kt = 0, ord = 0, next_tile = tile_of(1);
    RL_STAMP(0);

    for (int s = 0; tile < n_tiles; ++s) {
        // This wave's pieces of stage s have landed.  After a tile seam nothing is waited for: the step's loads were issued
        // before the epilogue, whose residual loads — younger, and returning in order — have been consumed since; waiting here
        // would wait for the acknowledgement of the epilogue's last stores.
        if (kt != 0 || ord == 0) __builtin_amdgcn_s_waitcnt(0x0f70);
        rl_barrier();   // ... and everyone's; every wave has left the other stage (K-step s - 1, or the epilogue's buffers)
        const bool seam = kt + 1 == nk;
        dma((s + 1) & 1, seam ? next_tile : tile, seam ? 0 : kt + 1);
        const _Float16* S = lds + (s & 1) * STAGE_H;
        // fragment t: rows 16 t + l15 of this wave's rows, logical chunk plane * 4 + q4
        int fl_ = pope_lane_id();
        asm volatile("" : "+v"(fl_));
        const int fl15 = fl_ & 15, fq4 = fl_ >> 4;
        const int sw_hi = 8 * (fq4 ^ (fl15 & 7)), sw_lo = 8 * ((4 + fq4) ^ (fl15 & 7));
        const int a_row = (wm * WMR + fl15) * ROWB, w_row = (RM + wn * 96 + fl15) * ROWB;
        // ni-major: the A fragments (hi, lo: 8 x 4 registers) stay for the K-step, the W fragments stream through two at
        // a time (lo, hi of column block ni), each feeding 12 MFMAs.  Per accumulator the order of the partial
        // products is that of gemm_planes.hip (lo.hi, hi.lo, hi.hi): bit-identical sums.
        // Program order is pinned with scheduling fences: the scheduler otherwise hoists all 20 fragment reads to the
        // top of the K-step (80 live registers).
        f16x8 ah[NMI], al[NMI], wl[2], wh[2];
        // issue order = order of first use (the LDS returns in order): the first MFMA (wl[0] . ah[0]) waits for two reads, every
        // further one of the first column block for one more — not for the whole 2 NMI + 2 (the waves of the workgroup all read at
        // this point: 14 reads x 8 waves are 900 cycles of the LDS array)
        wl[0] = *reinterpret_cast<const f16x8*>(S + w_row + sw_lo);
#pragma unroll
        for (int t = 0; t < NMI; ++t) ah[t] = *reinterpret_cast<const f16x8*>(S + a_row + t * 16 * ROWB + sw_hi);
        wh[0] = *reinterpret_cast<const f16x8*>(S + w_row + sw_hi);
#pragma unroll
        for (int t = 0; t < NMI; ++t) al[t] = *reinterpret_cast<const f16x8*>(S + a_row + t * 16 * ROWB + sw_lo);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) {
            const int cur = ni & 1, nxt = cur ^ 1;
            if (ni + 1 < 6) {
                wl[nxt] = *reinterpret_cast<const f16x8*>(S + w_row + (ni + 1) * 16 * ROWB + sw_lo);
                wh[nxt] = *reinterpret_cast<const f16x8*>(S + w_row + (ni + 1) * 16 * ROWB + sw_hi);
            }
            if constexpr (LN && !RES_TABLE)
                if (ni == 2 && g.rl_prefetch && kt >= nk - 1 - G::NPF && kt < nk - 1 && nk > G::NPF) prefetch_res(tile, kt - (nk - 1 - G::NPF));
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) acc[mi][ni] = mfma16(wl[cur], ah[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) acc[mi][ni] = mfma16(wh[cur], al[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) acc[mi][ni] = mfma16(wh[cur], ah[mi], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++kt == nk) {
            rl_barrier();   // every wave has finished reading stage s & 1: it holds the transposition buffers now
            if constexpr (LN) epilogue(tile, s & 1);
            else epilogue_planes(tile, s & 1);
            zero_acc();
            kt = 0;
            tile = next_tile;
            next_tile = tile_of(++ord + 1);
            rl_si += 8;
            RL_STAMP(0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // the last (re-)load into LDS has landed before the workgroup gives the LDS back
}

template <class G, int MODE, bool RES_TABLE>
int launch_rowln_geo(const GemmParams& g, hipStream_t stream) {
    static pope_dev_mask lds_ok{0};   // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_rowln16_kernel<G, MODE, RES_TABLE>, G::LDS_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    const int ncol = (g.N + RN - 1) / RN;
    const int tiles = ((g.M + G::RM - 1) / G::RM) * ncol, cus = pope_cu_count();
    hipLaunchKernelGGL((gemm_rowln16_kernel<G, MODE, RES_TABLE>), dim3(tiles < cus ? tiles : cus), dim3(G::RTH), G::LDS_BYTES, stream,
                       g, tiles, ncol);
    return pope_check_launch();
}

// Two tile geometries, the same arithmetic (bit-identical results): 192 rows (2 x 4 waves of 96 x 96: 1.0 staged byte per output
// and K-step) where its rounds over the CUs cost no more row-rounds than the 128-row tile's (2 x 4 waves of 64 x 96: 1.33 bytes),
// e.g. the 64-image chunk: 511 tiles = 2 rounds x 192 against 766 tiles = 3 rounds x 128; mid-size batches that fill less than a
// round of 192-row tiles keep the smaller tile.
template <int MODE, bool RES_TABLE>
int launch_rowln(const GemmParams& g, hipStream_t stream) {
    const long long cus = pope_cu_count(), ncol = (g.N + RN - 1) / RN;
    const long long r128 = (((g.M + 127) / 128) * ncol + cus - 1) / cus * 128, r192 = (((g.M + 191) / 192) * ncol + cus - 1) / cus * 192;
    return r192 <= r128 ? launch_rowln_geo<RlGeo<192, 2>, MODE, RES_TABLE>(g, stream)
                        : launch_rowln_geo<RlGeo<128, 2>, MODE, RES_TABLE>(g, stream);
}

}  // namespace

#ifdef RL_STAMPS
extern "C" int pope_lab_rowln_stamps(unsigned long long* host192) {
    return hipMemcpyFromSymbol(host192, HIP_SYMBOL(g_rl_dbg), sizeof(unsigned long long) * 192) == hipSuccess ? 0 : -1;
}
#endif

bool pope_gemm_rowln_supported(const GemmParams& g) {
    return g.N == RN && g.ldc == RN && g.K >= 2 * RK && (g.K % RK) == 0 && g.lda == g.K && g.ldw == g.K && g.ldres == RN &&
           size_t(g.M + 192) * g.lda * 4 < (size_t(1) << 32) && size_t(g.M + 192) * RN * 4 < (size_t(1) << 32) - 512;
}

// x = res + gamma * (A.W^T + bias) -> g.C (fp32, may alias res), LayerNorm(x; ln_w, ln_b, ln_eps) -> g.ln_planes or g.ln_f32
int pope_launch_gemm_rowln(const GemmParams& g_in, hipStream_t stream) {
    GemmParams g = g_in;
    g.rl_prefetch = 1;   // touch the tile's residual lines during its K loop (finding 21: proj -3 %)
    if (!g.a_pl || !g.w_pl || !g.C || !g.res || !g.ln_w || !g.ln_b || (!g.ln_planes) == (!g.ln_f32) || g.M <= 0) return POPE_ERR_ARG;
    if (!pope_gemm_rowln_supported(g)) return POPE_ERR_ARG;
    if (g.res_mod < 0 || (g.res_mod == 0 && !g.gamma)) return POPE_ERR_ARG;
    if (g.res_mod > 0) return g.ln_planes ? launch_rowln<RL_LN_PLANES, true>(g, stream) : launch_rowln<RL_LN_F32, true>(g, stream);
    return g.ln_planes ? launch_rowln<RL_LN_PLANES, false>(g, stream) : launch_rowln<RL_LN_F32, false>(g, stream);
}

// planes -> planes Linear (BIAS, BIAS_GELU) on the same stream for widths that are multiples of 384 (ViT-S/14: QKV 1 152, FC1
// 1 536) at batches of several rounds: no partial column tile, the next tile's first K-step lands under the epilogue
bool pope_stream384_supported(const GemmParams& g) {
    if (g.plain || !g.a_pl || !g.w_pl || !g.c_pl || g.conv_cch > 0 || g.nbatch > 1) return false;
    if (g.epilogue != EPI_BIAS && g.epilogue != EPI_BIAS_GELU) return false;
    if (g.N % RN || g.K < 2 * RK || (g.K % RK) || g.lda != g.K || g.ldw != g.K || g.ldc != g.N) return false;
    if (size_t(g.M + 192) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N) * g.ldw * 4 >= (size_t(1) << 32) ||
        size_t(g.M + 192) * g.ldc * 4 >= (size_t(1) << 32) - 512)
        return false;
    // several rounds of 192-row tiles: one workgroup per CU, nothing runs under a tile's K loop but its own staging
    return size_t((g.M + 191) / 192) * (g.N / RN) >= size_t(4) * pope_cu_count();
}

int pope_launch_stream384(const GemmParams& g, hipStream_t stream) {
    if (!pope_stream384_supported(g)) return POPE_ERR_ARG;
    return g.epilogue == EPI_BIAS_GELU ? launch_rowln_geo<RlGeo<192, 2>, RL_GELU_PLANES, false>(g, stream)
                                       : launch_rowln_geo<RlGeo<192, 2>, RL_BIAS_PLANES, false>(g, stream);
}
