// f16x3 "planes" NT GEMM on v_mfma_f32_16x16x32_f16: the mainloop of every Linear layer, of the LoFTR convolutions and of
// the dense matcher's contraction.  Contract: operands are f16 hi/lo planes written once by their producer (LayerNorm, a GELU
// epilogue, the weight loader), so the K loop is loads -> ds_write_b128 -> ds_read_b128 -> MFMA with no VALU work; persistent
// tile stream; LDS-transposed coalesced epilogues.  Why this MFMA shape (round 2; a 32x32x16 twin of this kernel existed
// until round 4 as an A/B reference and was deleted with the other dev switches):
//   * measured on this chip (scripts/mfma_shape_lab2.hip: this K-step on random operands re-read from LDS): the
//     16x16x32 form needs the same cycles per FLOP but the chip holds 1.90 GHz under it instead of 1.67 GHz under
//     the 32x32x16 form -> 1 871 vs 1 651 TFLOP/s executed (+13 %); MI355X_MICROARCH.md "DVFS give-back" item 7;
//   * its C^T accumulator block gives a lane FOUR CONSECUTIVE output columns of one row, so the epilogue's LDS
//     transposition writes 16-byte pieces.
// A wave owns 64 x 64 of the 128 x 128 tile as 4 x 4 blocks of 16 x 16 (64 accumulator registers); one
// K-step (32) is ONE k-step of the MFMA: 16 fragment reads (ds_read_b128) feed 48 MFMAs (16 per partial product).
// LDS rows are 160 bytes ([32 hi | 32 lo] halves + 32 B pad): conflict-free for the 16-row x 4-chunk fragment
// reads; two stages x 256 rows = 80 KB per workgroup, two workgroups per CU = all 160 KB.
#include "gemm_core.h"
#include "kernels.h"

namespace {

using namespace gemm_core;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int ROWH = 80;                     // halves per LDS row
constexpr int OPER = 128 * ROWH;             // halves per operand tile
constexpr int STAGE = 2 * OPER;              // A rows, then W rows
constexpr size_t P16_LDS_BYTES = size_t(2) * STAGE * sizeof(_Float16);   // 81 920
static_assert(size_t(STAGE) * sizeof(_Float16) >= size_t(4) * 32 * EPI_ST * sizeof(float), "epilogue staging must fit a stage");
constexpr float A_SCALE = K_PLANES_ACT_SCALE, W_SCALE = K_PLANES_W_SCALE;
constexpr float L2E = 1.44269504088896340736f;

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// exact-erf GELU on a pair, Abramowitz-Stegun 7.1.26 form (derivation: gemm_f32.hip:gelu_erf2)
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

// PLAIN (GemmParams::plain): single-product f16 arithmetic on the same loader, LDS layout and epilogues.  An f16
// row-major tensor [rows, 2 ld] IS a planes tensor [rows, ld] whose "lo" half of a 128-byte chunk holds the NEXT 32
// columns instead of the residuals of the first 32; a K-step then covers 64 real columns with 32 MFMAs (hi.hi + lo.lo)
// instead of 32 columns with 48.  K, lda, ldw (and ldc of a planes output) count 64-bit column pairs, N / bias /
// residual stay in real columns; a planes output is written as f16 row-major (value * 8).  BASELINE config 5's "fp16".
template <int EPI, bool OUT_PLANES, bool CONV = false, bool PLAIN = false>
__global__ __launch_bounds__(THREADS, 2) void gemm_planes16_kernel(const GemmParams g, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);
    constexpr int NLD = 8;   // 16-byte pieces per thread and K-step: 4 A rows + 4 W rows

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_pb = ((g.M + BM - 1) / BM) * tiles_n;  // tiles per batch (EPI_SIM)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, q4 = lane >> 4;   // fragment / accumulator coordinates of the 16x16x32 MFMA
    // Staging: a row's K-step is 128 contiguous bytes in memory ([32 hi | 32 lo] halves) = eight 16-byte pieces = eight
    // consecutive lanes -> whole cache lines, and the same 128 contiguous bytes in LDS.  Thread -> rows prow + 32 i, piece pc.
    const int prow = tid >> 3, pc = tid & 7;
    const int nk = g.K / BK;  // >= 2 (launcher)
    const unsigned nb = EPI == EPI_SIM ? unsigned(g.nbatch) : 1u;
    const unsigned a_rows = CONV ? unsigned(g.M) + 2u * unsigned(g.conv_wp) + 2u : nb * unsigned(g.M);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a_pl), 0, a_rows * unsigned(g.lda) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_pl), 0, nb * unsigned(g.N) * unsigned(g.ldw) * 4u, 0x00020000);
    // CONV: byte offset of the next K-step's A rows = (dy * Wp + dx) rows + chunk * 128, kept incrementally (scalar
    // selects: the K-step stays one basic block)
    const int cv_row = g.lda * 4, cv_dy = (g.conv_wp - 2) * cv_row;
    // K order CHUNK-major: the nine taps of a 32-channel chunk back to back, so that the three tap rows — three different
    // image rows, each also read by the tiles above and below — are touched within nine K-steps of each other instead of
    // 3 * chunks apart.  Tap-major order let the 4 MB L2 of an XCD turn over in between: PMC fetch 3.2x (128 channels) to
    // 8.5x (224) the input, 38.6 GB per 48-image ResNet-FPN call against 22.6 GB now (profiles/r03/pmc_cnn.json).  W's K
    // offset (tap * chunks + chunk: the folded matrices keep their tap-major columns) is its own counter.
    int cv_chunk = 0, cv_dx = 0, cv_off = 0, cv_dyi = 0, cv_w = 0;

    // Tile stream of this persistent workgroup (DESIGN.md findings 5-7 have the measurements behind every choice here): full
    // rounds by XCD-remapped id, the partial last round one tile per CU by raw blockIdx; the K-steps of consecutive
    // tiles are ONE stream (item = (tile, kt)) through the double-buffered LDS, two register sets of loads in flight.
    // The bookkeeping uses asm selects: a K-step must stay ONE basic block for the pinned instruction mix.
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
    u32x4 r0[NLD], r1[NLD];  // A rows, then W rows
    int ld_ord = 0, ord = 0;
    int ld_tile = first, ld_kt = 0;  // next stream item to load
    auto load_next = [&](u32x4 (&st)[NLD]) {
        // past the end of the stream the last tile is re-loaded and never consumed
        const int lt = ld_tile < n_tiles ? ld_tile : n_tiles - 1;
        int m0, n0;
        if constexpr (EPI == EPI_SIM) {  // batched: rows of batch b start at b * M (A) / b * N (W)
            const int b = lt / tiles_pb, rem = lt - b * tiles_pb;
            m0 = b * g.M + (rem / tiles_n) * BM;
            n0 = b * g.N + (rem % tiles_n) * BN;
        } else {
            m0 = (lt / tiles_n) * BM;
            n0 = (lt % tiles_n) * BN;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned va = unsigned(m0 + prow + 32 * i) * unsigned(g.lda) * 4u + pc * 16u;
            st[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, va, CONV ? cv_off + cv_chunk * 128 : ld_kt * 128, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vw = unsigned(n0 + prow + 32 * i) * unsigned(g.ldw) * 4u + pc * 16u;
            st[4 + i] = __builtin_amdgcn_raw_buffer_load_b128(rw, vw, CONV ? cv_w : ld_kt * 128, 0);
        }
        const int wrap = ++ld_kt == nk;
        if constexpr (CONV) {
            const int xw = cv_dx == 2;                                    // end of a tap row
            const int tw = xw & (cv_dyi == 2);                            // ninth tap done: next channel chunk
            cv_off += pope_uniform_select(xw, cv_dy, cv_row);
            cv_off = pope_uniform_select(tw, 0, cv_off);
            cv_dx = pope_uniform_select(xw, 0, cv_dx + 1);
            cv_dyi = pope_uniform_select(tw, 0, cv_dyi + xw);
            cv_chunk += tw;
            cv_w = pope_uniform_select(tw, cv_chunk * 128, cv_w + g.conv_cch * 128);   // W column block = tap * cch + chunk
            cv_chunk = pope_uniform_select(wrap, 0, cv_chunk);            // next tile
            cv_w = pope_uniform_select(wrap, 0, cv_w);
        }
        ld_kt = pope_uniform_select(wrap, 0, ld_kt);
        ld_ord += wrap;
        ld_tile = tile_of(ld_ord);
    };
    auto write_stage = [&](int s, const u32x4 (&st)[NLD]) {
        _Float16* S = lds + s * STAGE + 8 * pc;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(S + (prow + 32 * i) * ROWH) = st[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(S + OPER + (prow + 32 * i) * ROWH) = st[4 + i];
    };
    // fragment t of an operand: rows 16 t + l15 of this wave's 64, k chunk q4 (8 halves) of the hi / lo plane
    const int a_off = (wm * 64 + l15) * ROWH + 8 * q4, w_off = OPER + (wn * 64 + l15) * ROWH + 8 * q4;
    f32x4 acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    struct Frags { f16x8 ah[4], al[4], wh[4], wl[4]; };

    const unsigned c_row_bytes = unsigned(g.ldc) * 4u;  // fp32 rows and planes rows have the same pitch
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        OUT_PLANES ? g.c_pl : static_cast<void*>(g.C), 0, nb * unsigned(g.M) * c_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == EPI_BIAS_LS_RES ? g.res : g.C), 0,
        EPI == EPI_BIAS_LS_RES ? unsigned(g.res_mod > 0 ? g.res_mod : g.M) * unsigned(g.ldres) * 4u : 0u, 0x00020000);
    auto res_row = [&](unsigned row) -> unsigned { return g.res_mod > 0 ? row % unsigned(g.res_mod) : row; };
    const __amdgpu_buffer_rsrc_t rresp = __builtin_amdgcn_make_buffer_rsrc(   // EPI_CONV: residual as activation planes
        const_cast<void*>(EPI == EPI_CONV ? g.res_pl : nullptr), 0,
        EPI == EPI_CONV && g.res_pl ? unsigned(g.M) * unsigned(g.ldres_pl) * 4u : 0u, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rup = __builtin_amdgcn_make_buffer_rsrc(   // EPI_CONV: the FPN merge's half-resolution map
        const_cast<float*>(EPI == EPI_CONV_UP ? g.up_src : nullptr), 0,
        EPI == EPI_CONV_UP ? unsigned(g.up_n) * unsigned((g.up_hp - 2) / 2 + 2) * unsigned((g.up_wp - 2) / 2 + 2) * unsigned(g.up_lds) * 4u : 0u,
        0x00020000);
    const int ec4 = (lane & 15) * 4, elr = lane >> 4;   // row-layout coordinates after the LDS transposition
    constexpr unsigned DROP = 0xFFFFFF00u;              // beyond every buffer extent: the access is discarded

    // accumulator blocks of the 32-row half mh -> this wave's LDS staging rows (16-byte pieces: a lane holds four
    // consecutive columns of one row), conflict-free (row stride 68 floats: eight rows x 4 banks)
    auto stage_half = [&](float* E, int mh) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4*>(&E[(m2 * 16 + l15) * EPI_ST + ni * 16 + 4 * q4]) = acc[2 * mh + m2][ni];
    };

    // Epilogue of one tile: branch-free, no loads between its stores (bias / gamma hoisted; rows >= M and columns >= N
    // dropped by the buffer range check; residual rows fetched eight at a time): finding 7 of DESIGN.md.
    auto epilogue = [&](int tile, float* epi) {
        if constexpr (EPI == EPI_SIM) {
            // Similarity tile of batch b: sim = acc / divisor_eff (divisor_eff = T * 2^16: the operand scales are exact
            // powers of two), stored to C[b][row][col], plus this wave's partial softmax statistics of the tile — the
            // dual softmax of coarse_matching.py:119 needs max and sum(exp) of every row AND every column of sim, and
            // computing their per-tile pieces here, from registers, replaces two full passes over the L x S matrix.
            // Rows >= M and columns >= N belong to the next batch's operands (or the zero fill): they are set to -inf
            // right after the scaling, so they vanish from every maximum and every sum; their stores are dropped.
            const int b = tile / tiles_pb, rem = tile - b * tiles_pb;
            const int tm = rem / tiles_n, tn = rem - tm * tiles_n;
            const int m0s = tm * BM, n0s = tn * BN;
            const int cols = n0s + wn * 64 + ec4;
            // x / d with a reciprocal and one correction step (q = x r; e = x - d q (exact fma); q += e r): the correctly
            // rounded quotient for all but pathological divisors, 3 instructions instead of the ~10 of a full division
            const float dv = g.divisor_eff, rdiv = g.rdiv;
            const bool edge = m0s + BM > g.M || n0s + BN > g.N;   // wave-uniform
            // pin the epilogue arithmetic behind the tile-end branch (else it is speculated into every K-step)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) asm volatile("" : "+v"(acc[mi][ni]));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = acc[mi][ni][j];
                        float q = x * rdiv;
                        q = __builtin_fmaf(__builtin_fmaf(-dv, q, x), rdiv, q);
                        acc[mi][ni][j] = q;
                    }
            if (edge) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const bool row_out = m0s + wm * 64 + mi * 16 + l15 >= g.M;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (row_out || n0s + wn * 64 + ni * 16 + 4 * q4 + j >= g.N) acc[mi][ni][j] = -INFINITY;
                }
            }
            // ---- row statistics over this wave's 64 columns, in the accumulator layout: the four lanes l15 + 16 q
            // hold 16 values each of row (mi, l15)
            if (g.row_part) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    float m = -INFINITY;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int j = 0; j < 4; ++j) m = __builtin_fmaxf(m, acc[mi][ni][j]);
                    float a, c;
                    pope_xor16_pair(m, a, c);
                    pope_xor32_pair(__builtin_fmaxf(a, c), a, c);
                    m = __builtin_fmaxf(a, c);
                    const float ms = m == -INFINITY ? 0.f : m;   // an all-padding block contributes (max -inf, sum 0)
                    f32x2 sum = {0.f, 0.f};
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int j = 0; j < 4; j += 2)
                            sum += f32x2{__builtin_amdgcn_exp2f((acc[mi][ni][j] - ms) * L2E),
                                         __builtin_amdgcn_exp2f((acc[mi][ni][j + 1] - ms) * L2E)};
                    pope_xor16_pair(sum[0] + sum[1], a, c);
                    pope_xor32_pair(a + c, a, c);
                    const int row = m0s + wm * 64 + mi * 16 + l15;
                    if (q4 == 0 && row < g.M)
                        *reinterpret_cast<f32x2*>(g.row_part + ((size_t(b) * g.M + row) * g.ncb + tn * 2 + wn) * 2) = f32x2{m, a + c};
                }
            }
            __syncthreads();
            float* Es = epi + wave * 32 * EPI_ST;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                // ---- transposition to rows, coalesced store of sim, column statistics over this block's 32 rows
                stage_half(Es, mh);
                __builtin_amdgcn_wave_barrier();
                f32x4 vr[8];
                f32x4 cm = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 v = vr[i] = *reinterpret_cast<const f32x4*>(&Es[(elr + 4 * i) * EPI_ST + ec4]);
                    const int row = m0s + wm * 64 + mh * 32 + elr + 4 * i;
#pragma unroll
                    for (int e = 0; e < 4; ++e) cm[e] = __builtin_fmaxf(cm[e], v[e]);
                    const unsigned off = (unsigned(b) * unsigned(g.M) + unsigned(row)) * c_row_bytes + unsigned(cols) * 4u;
                    const bool row_ok = row < g.M;
                    if (!(g.N & 1)) {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v[0], v[1]}), rc,
                                                              row_ok && cols + 1 < g.N ? off : DROP, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v[2], v[3]}), rc,
                                                              row_ok && cols + 3 < g.N ? off + 8u : DROP, 0, 2);
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
                        const size_t o = (size_t(b) * g.nrb + tm * 4 + wm * 2 + mh) * g.ldp + cols;
                        *reinterpret_cast<f32x4*>(g.col_pmax + o) = cm;
                        *reinterpret_cast<f32x4*>(g.col_psum + o) = cs;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            return;
        }
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const int col = n0 + wn * 64 + ec4;
        const bool col_ok = col < g.N;
        // EPI_CONV planes rows are wider than N when the channel count is not a multiple of 32 (196 -> 224): the
        // padding columns are the next convolution's K range and must read as zeros
        const bool col_pad = (EPI == EPI_CONV || EPI == EPI_CONV_UP) && OUT_PLANES && !col_ok && col < g.ldc;
        const int colc = col_ok ? col : 0;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f}, gamma = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias = *reinterpret_cast<const f32x4*>(g.bias + colc);
        if constexpr (EPI == EPI_BIAS_LS_RES) gamma = g.gamma ? *reinterpret_cast<const f32x4*>(g.gamma + colc) : f32x4{1.f, 1.f, 1.f, 1.f};
        constexpr float inv = 1.0f / (A_SCALE * W_SCALE);
        if constexpr (EPI == EPI_BIAS_LS_RES) {  // res + (v*inv + bias)*gamma = res + v*(inv*gamma) + bias*gamma
            bias = bias * gamma;
            gamma = gamma * inv;
        }
        // EPI_SAM_QKV: this wave's 64 columns lie in ONE of q / k / v (dim % 64 == 0); a lane's four columns in one head
        [[maybe_unused]] int sq_head = 0, sq_c = 0, sq_row_h = 0, sq_lo = 0;
        [[maybe_unused]] float sq_scale = 1.0f;
        [[maybe_unused]] __amdgpu_buffer_rsrc_t rsq = rres, rmap = rres;
        if constexpr (EPI == EPI_SAM_QKV) {
            const int wcol = n0 + wn * 64;
            const int which = __builtin_amdgcn_readfirstlane((wcol < g.N ? wcol : 0) / g.sam_dim);
            const int rem = colc - which * g.sam_dim;
            sq_head = rem / g.sam_hd;
            sq_c = rem - sq_head * g.sam_hd;
            // rows: Q' [DQ hi | DQ lo], K' [DQ hi | hd lo], V [DV hi | DV lo]; PLAIN: the hi parts only
            sq_row_h = which == 0 ? (PLAIN ? g.sam_dq : 2 * g.sam_dq) : which == 1 ? (PLAIN ? g.sam_dq : g.sam_dq + g.sam_hd)
                                                                                   : (PLAIN ? g.sam_dv : 2 * g.sam_dv);
            sq_lo = which == 2 ? g.sam_dv : g.sam_dq;
            sq_scale = which == 0 ? g.sam_qscale : 1.0f;
            void* dst = which == 0 ? g.sam_q : which == 1 ? g.sam_k : g.sam_v;
            const unsigned bytes = which == 0 ? g.sam_bytes[0] : which == 1 ? g.sam_bytes[1] : g.sam_bytes[2];
            rsq = __builtin_amdgcn_make_buffer_rsrc(dst, 0, bytes, 0x00020000);
            rmap = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(g.sam_rowmap), 0, unsigned(g.M) * 4u, 0x00020000);
        }
        [[maybe_unused]] float qkv_scale = 1.0f;
        if constexpr (EPI == EPI_QKV_F16) qkv_scale = colc < g.sam_dim ? g.sam_qscale : 1.0f;   // a lane's four columns lie in one of q / k / v
        __syncthreads();  // all waves have finished reading the last K-step stage
        float* E = epi + wave * 32 * EPI_ST;
        f32x2 amax = {0.f, 0.f};  // OUT_PLANES: largest magnitude written as planes (range guard; rows >= M hold finite
                                  // junk computed from zero-filled operands: bias / gelu(bias), as real rows see)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            stage_half(E, mh);
            const unsigned row0 = unsigned(m0 + wm * 64 + mh * 32 + elr);
            f32x4 res[8];
            if constexpr (EPI == EPI_BIAS_LS_RES) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 rres, col_ok ? res_row(row0 + 4 * i) * unsigned(g.ldres) * 4u + unsigned(col) * 4u : DROP, 0, 0));
            }
            [[maybe_unused]] unsigned sq_dest[8];
            if constexpr (EPI == EPI_SAM_QKV) {   // destination row of each token row: (window batch, head 0) base + token-in-window
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    sq_dest[i] = __builtin_amdgcn_raw_buffer_load_b32(rmap, (row0 + 4 * i) * 4u, 0, 0) + unsigned(sq_head * g.sam_npad);
            }
            if constexpr (EPI == EPI_CONV) {   // shortcut rows: (hi + lo) / 8; an empty descriptor (no residual) reads zeros
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned o = (row0 + 4 * i) * unsigned(g.ldres_pl) * 4u + unsigned((col >> 5) * 128 + (col & 31) * 2);
                    const f16x4 rh = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rresp, col_ok ? o : DROP, 0, 0));
                    const f16x4 rl = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rresp, col_ok ? o + 64u : DROP, 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) res[i][e] = (float(rh[e]) + float(rl[e])) * (1.0f / A_SCALE);
                }
            }
            // EPI_CONV_UP — FPN merge: the bilinear x2 (align_corners) sample of the half-resolution map at each row's pixel.  Pixel of
            // the pass's first row by multiply-high with host-made reciprocals + one correction (a device-side `/` keeps its
            // reciprocal in VGPRs across the K loop), of the following rows (+ 4 each) by carries.
            [[maybe_unused]] unsigned pb = 0, pyp = 0, pxp = 0;
            if constexpr (EPI == EPI_CONV_UP) {
                const unsigned hpwp = unsigned(g.up_hp) * unsigned(g.up_wp);
                pb = __umulhi(row0, g.up_m_hw);
                unsigned prem = row0 - pb * hpwp;
                if (prem >= hpwp) { ++pb; prem -= hpwp; }
                pyp = __umulhi(prem, g.up_m_w);
                pxp = prem - pyp * unsigned(g.up_wp);
                if (pxp >= unsigned(g.up_wp)) { ++pyp; pxp -= unsigned(g.up_wp); }
            }
            [[maybe_unused]] auto up_rows = [&](int i_lo, int i_hi) __attribute__((always_inline)) {
                const int Hp = g.up_hp, Wp = g.up_wp, H = Hp - 2, W = Wp - 2, Hs = H / 2, Ws = W / 2, Wsp = Ws + 2, Hsp = Hs + 2;
                const float sh = g.up_sh, sw = g.up_sw;
#pragma unroll
                for (int i = i_lo; i < i_hi; ++i) {
                    const bool inside = col_ok && row0 + 4 * i < unsigned(g.M) && pyp >= 1u && pyp <= unsigned(H) && pxp >= 1u && pxp <= unsigned(W);
                    const int y = inside ? int(pyp) - 1 : 0, x = inside ? int(pxp) - 1 : 0;
                    const float ry = sh * float(y), rx = sw * float(x);
                    const int y0 = int(ry), x0 = int(rx);
                    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
                    const float ly1 = ry - float(y0), lx1 = rx - float(x0), ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                    const unsigned base = (pb * unsigned(Hsp) + 1u) * unsigned(Wsp) + 1u;
                    auto at = [&](int yy, int xx) {
                        const unsigned o = ((base + unsigned(yy) * unsigned(Wsp) + unsigned(xx)) * unsigned(g.up_lds) + unsigned(col)) * 4u;
                        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rup, inside ? o : DROP, 0, 0));
                    };
                    const f32x4 v00 = at(y0, x0), v01 = at(y0, x1), v10 = at(y1, x0), v11 = at(y1, x1);
#pragma unroll
                    for (int e = 0; e < 4; ++e)   // conv.hip:upsample_add_planes_kernel's expression, term for term
                        res[i][e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
                    pxp += 4u;                                        // Wp >= 10: at most one carry each
                    const bool cx = pxp >= unsigned(Wp);
                    pxp -= cx ? unsigned(Wp) : 0u;
                    pyp += cx ? 1u : 0u;
                    const bool cy = pyp >= unsigned(Hp);
                    pyp -= cy ? unsigned(Hp) : 0u;
                    pb += cy ? 1u : 0u;
                }
            };
            __builtin_amdgcn_wave_barrier();
            auto store_rows = [&](int i_lo, int i_hi) __attribute__((always_inline)) {
#pragma unroll
            for (int i = i_lo; i < i_hi; ++i) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&E[(elr + 4 * i) * EPI_ST + ec4]);
                const unsigned off = (row0 + 4 * i) * c_row_bytes;
                if constexpr (EPI == EPI_BIAS) {
                    v = v * inv + bias;
                } else if constexpr (EPI == EPI_SAM_QKV) {
                    v = (v * inv + bias) * sq_scale;
                    pope_amax4x2(amax, v);
                    const unsigned o = col_ok && row0 + 4 * i < unsigned(g.M) ? (sq_dest[i] * unsigned(sq_row_h) + unsigned(sq_c)) * 2u : DROP;
                    if constexpr (PLAIN) {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4)), rsq, o, 0, 0);
                    } else {
                        f16x4 hi, lo;
                        pope_split4(v, hi, lo);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rsq, o, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rsq, o + unsigned(sq_lo) * 2u, 0, 0);
                    }
                    continue;
                } else if constexpr (EPI == EPI_QKV_F16) {   // attention operands: f16 row-major, no activation scale
                    v = (v * inv + bias) * qkv_scale;
                    pope_amax4x2(amax, v);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4)), rc,
                                                          col_ok ? off + unsigned(col) * 2u : DROP, 0, 0);
                    continue;
                } else if constexpr (EPI == EPI_BIAS_GELU) {
                    v = v * inv + bias;
                    const f32x2 g01 = gelu_erf_pair(f32x2{v[0], v[1]}), g23 = gelu_erf_pair(f32x2{v[2], v[3]});
                    v = f32x4{g01[0], g01[1], g23[0], g23[1]};
                } else if constexpr (EPI == EPI_BIAS_RELU) {
                    v = v * inv + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(v[e], 0.f);
                } else if constexpr (EPI == EPI_CONV || EPI == EPI_CONV_UP) {
                    v = (v * inv + bias) + res[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(v[e], 0.f) + g.act_slope * __builtin_fminf(v[e], 0.f);
                } else {
                    v = res[i] + v * gamma + bias;
                }
                if constexpr (OUT_PLANES && PLAIN) {   // f16 row-major, value * 8
                    pope_amax4x2(amax, v);
                    const f16x4 h = __builtin_convertvector(v * A_SCALE, f16x4);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rc, col_ok ? off + unsigned(col) * 2u : DROP, 0, 2);
                } else if constexpr (OUT_PLANES) {
                    f16x4 hi, lo;
                    pope_amax4x2(amax, v);
                    pope_split4(v * A_SCALE, hi, lo);
                    if constexpr (EPI == EPI_CONV || EPI == EPI_CONV_UP) {
                        if (col_pad) { hi = f16x4{0, 0, 0, 0}; lo = f16x4{0, 0, 0, 0}; }
                    }
                    // planes row: per 32-column chunk [32 hi | 32 lo] halves
                    const unsigned o = col_ok || col_pad ? off + unsigned((col >> 5) * 128 + (col & 31) * 2) : DROP;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rc, o, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rc, o + 64u, 0, 2);
                } else {
                    // write-once outputs are stored non-temporally (they must not displace the A/W panels in L2); the
                    // residual stream (LS_RES) is re-read by the next LayerNorm and keeps the default policy
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc,
                                                           col_ok ? off + unsigned(col) * 4u : DROP, 0,
                                                           EPI == EPI_BIAS_LS_RES ? 0 : 2);
                }
            }
            };
            if constexpr (EPI == EPI_CONV_UP) {   // four rows at a time: the persistent mainloop keeps two K-steps of staging registers
                up_rows(0, 4);                    // alive across the epilogue, and 8 rows of neighbours on top of them spill those
                store_rows(0, 4);
                up_rows(4, 8);
                store_rows(4, 8);
            } else {
                store_rows(0, 8);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (EPI == EPI_SAM_QKV || EPI == EPI_QKV_F16)   // attention operands carry no scale
            pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) < POPE_F16_OVERFLOW));
        else if constexpr (OUT_PLANES)
            pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) * A_SCALE < POPE_F16_OVERFLOW));
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

    // One stream item: `nx` holds item s+1 (published to the other stage), then is refilled with item s+3.  The lo-of-W
    // and hi-of-A fragments (first partial product) are read first; the staging work and the second batch of reads are
    // pinned between the MFMAs (LLVM sched groups 0x8 MFMA, 0x100 DS read, 0x200 DS write, 0x20 VMEM read).
    auto item = [&](int s, u32x4 (&nx)[NLD]) {
        Frags f;
        const _Float16* S = lds + (s & 1) * STAGE;
        // first batch of fragment reads = the operands of the first product (f16x3: W lo, A hi; PLAIN: W hi, A hi)
        constexpr int W_FIRST = PLAIN ? 0 : 32, W_SECOND = PLAIN ? 32 : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            (PLAIN ? f.wh[t] : f.wl[t]) = *reinterpret_cast<const f16x8*>(S + W_FIRST + w_off + t * 16 * ROWH);
            f.ah[t] = *reinterpret_cast<const f16x8*>(S + a_off + t * 16 * ROWH);
        }
        write_stage((s + 1) & 1, nx);
        load_next(nx);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            (PLAIN ? f.wl[t] : f.wh[t]) = *reinterpret_cast<const f16x8*>(S + W_SECOND + w_off + t * 16 * ROWH);
            f.al[t] = *reinterpret_cast<const f16x8*>(S + 32 + a_off + t * 16 * ROWH);
        }
        // accumulators hold C^T (A-operand = W fragment, B-operand = A fragment); small terms first; term-major: an
        // accumulator is touched again only 16 MFMAs later
        if constexpr (PLAIN) {   // "lo" = the second 32 columns of the K-step: two plain products
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(f.wh[ni], f.ah[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(f.wl[ni], f.al[mi], acc[mi][ni]);
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(f.wl[ni], f.ah[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(f.wh[ni], f.al[mi], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(f.wh[ni], f.ah[mi], acc[mi][ni]);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // wl, ah
#pragma unroll
        for (int i = 0; i < (PLAIN ? 32 : 48); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NLD) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (i >= 4 && i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // wh, al
            if (i >= 14 && i < 14 + NLD) {   // one buffer load (and its two address instructions) per MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __syncthreads();  // stage (s+1)&1 is published, stage s&1 is free
        if (++kt == nk) {
            epilogue(tile, reinterpret_cast<float*>(lds + (s & 1) * STAGE));
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

template <int EPI, bool OUT_PLANES, bool CONV = false, bool PLAIN = false>
int launch16(const GemmParams& g, hipStream_t stream, int nbatch = 1) {
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_planes16_kernel<EPI, OUT_PLANES, CONV, PLAIN>, P16_LDS_BYTES, lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = nbatch * ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const int slots = 2 * pope_cu_count();   // two resident workgroups per CU (2 x 80 KB LDS)
    hipLaunchKernelGGL((gemm_planes16_kernel<EPI, OUT_PLANES, CONV, PLAIN>), dim3(tiles < slots ? tiles : slots), dim3(THREADS), P16_LDS_BYTES,
                       stream, g, tiles);
    return pope_check_launch();
}

}  // namespace

// argument checks are the callers' (gemm_f16x3.hip: pope_launch_gemm_nt_f16x3_planes / pope_launch_sim_f16x3_planes)
int pope_launch_planes16(const GemmParams& g, hipStream_t stream) {
    const bool out_planes = g.c_pl != nullptr;
    if (g.epilogue == EPI_SAM_QKV) {
        if (!g.sam_q || !g.sam_k || !g.sam_v || !g.sam_rowmap || g.sam_hd <= 0 || (g.sam_hd & 3) || (g.sam_dim & 63) || g.N != 3 * g.sam_dim)
            return POPE_ERR_ARG;
        if (g.plain && pope_plain256_supported(g)) return pope_launch_plain256(g, stream);
        return g.plain ? launch16<EPI_SAM_QKV, true, false, true>(g, stream) : launch16<EPI_SAM_QKV, true, false, false>(g, stream);
    }
    if (g.plain && pope_plain256_supported(g)) return pope_launch_plain256(g, stream);   // long-K shapes at large M: gemm_plain.hip
    if (g.plain) {   // single-product f16 (SAM encoder, precision "f16"): the four forms that path uses
        if (g.epilogue == EPI_BIAS && !out_planes) return launch16<EPI_BIAS, false, false, true>(g, stream);
        if (g.epilogue == EPI_BIAS_GELU && out_planes) return launch16<EPI_BIAS_GELU, true, false, true>(g, stream);
        if (g.epilogue == EPI_QKV_F16 && out_planes) return launch16<EPI_QKV_F16, true, false, true>(g, stream);
        if (g.epilogue == EPI_BIAS_LS_RES && !out_planes) return launch16<EPI_BIAS_LS_RES, false, false, true>(g, stream);
        if (g.epilogue == EPI_CONV && !out_planes && g.conv_cch > 0 && g.K == 9 * 32 * g.conv_cch && g.lda == 32 * g.conv_cch &&
            g.conv_wp >= 3)
            return launch16<EPI_CONV, false, true, true>(g, stream);
        return POPE_ERR_ARG;
    }
    // planes -> planes at large M (QKV, FC1 of the ViT blocks): the 256 x 256 LDS-direct mainloop of gemm_plain.hip, same bits
    // (round 4, same-box A/B inside bench.py: QKV 0.285 -> 0.273 ms, FC1 0.427 -> 0.351 ms, step 92.9 -> 89.0 ms)
    // large planes -> planes Linears on the LDS-direct mainloops (bit-identical to the tile kernel below): widths that leave a
    // partial 256-column tile but are multiples of 384 (QKV 1 152: 4.5 tiles of 256) on the 192 x 384 stream of gemm_rowln.hip
    // (0.294 -> 0.266 ms; FC1's 1 536 = 6 x 256 is 2 % faster on the 256 x 256 tiles: profiles/r04/stream384_ab.txt)
    if ((g.N & 255) && pope_stream384_supported(g)) return pope_launch_stream384(g, stream);
    if (pope_wide_x3_supported(g)) return pope_launch_wide_x3(g, stream);
    switch (g.epilogue) {
        case EPI_BIAS: return out_planes ? launch16<EPI_BIAS, true>(g, stream) : launch16<EPI_BIAS, false>(g, stream);
        case EPI_BIAS_GELU: return out_planes ? launch16<EPI_BIAS_GELU, true>(g, stream) : launch16<EPI_BIAS_GELU, false>(g, stream);
        case EPI_BIAS_RELU: return out_planes ? launch16<EPI_BIAS_RELU, true>(g, stream) : launch16<EPI_BIAS_RELU, false>(g, stream);
        case EPI_BIAS_LS_RES: return launch16<EPI_BIAS_LS_RES, false>(g, stream);
        case EPI_SIM: return launch16<EPI_SIM, false>(g, stream, g.nbatch);
        case EPI_CONV:
            if (g.conv_cch > 0) {
                if (g.K != 9 * 32 * g.conv_cch || g.lda != 32 * g.conv_cch || g.conv_wp < 3) return POPE_ERR_ARG;
                if (pope_wide_conv_supported(g)) return pope_launch_wide_conv(g, stream);   // 256-row LDS-direct tiles, same bits
                return out_planes ? launch16<EPI_CONV, true, true>(g, stream) : launch16<EPI_CONV, false, true>(g, stream);
            }
            if (g.up_src) return out_planes && !g.res_pl ? launch16<EPI_CONV_UP, true>(g, stream) : POPE_ERR_ARG;
            return out_planes ? launch16<EPI_CONV, true>(g, stream) : launch16<EPI_CONV, false>(g, stream);
    }
    return POPE_ERR_ARG;
}
