// Plain-f16 NT GEMM for the long-K Linear layers of BASELINE config 5 (POPE_PREC_F16: SAM ViT-H, DINOv2 ViT-L/14 — one MFMA
// per product, f16 row-major operands with value * 8 / value * 256, fp32 accumulate).
//
// Why a second mainloop (round 4).  Until now these GEMMs ran on gemm_planes16_kernel's PLAIN flag ("an f16 row-major tensor
// IS a planes tensor"): 128 x 128 tiles, both operands staged through VGPRs into LDS.  With one MFMA per product that tile
// asks the CU for 64 B/clk of L2 -> LDS traffic at the full matrix rate — the whole vector-memory path — and ran at 0.29-0.31
// of the f16 peak (QKV 780, FC1 725 TFLOP/s on ViT-H at 4 images; the vendor's f16 GEMM on the same shapes: 1 050 / 1 140,
// profiles/r04/vendor_gemm_probe_config5.txt).  Here:
//   * 256 x 256 x 64 tiles (256 x 128 for N < 512), 8 waves, one workgroup per CU: half the operand bytes per MFMA;
//   * operands go memory -> LDS directly (buffer_load ... lds, 1 KB per wave-instruction, no VGPRs, no ds_write: the
//     VGPR -> LDS store path is what the f16x3 kernels' staging is bound by, DESIGN.md finding 6); unpadded 128-byte rows,
//     16-byte piece p of row r at position p ^ (r & 7) (applied on the source side) -> conflict-free 16-row fragment reads;
//     two 64 KB stages (three 48 KB stages for the 128-column tiles), one barrier per K-step;
//   * a wave owns 128 x 64 (two 64 x 64 sub-tiles that share their W fragments): 24 fragment reads feed 64 MFMAs per K-step
//     (0.375 per MFMA; the 128 x 128 kernel in PLAIN mode: 0.5);
//   * the epilogues are gemm_planes.hip's, sub-tile by sub-tile (LDS-transposed, coalesced, the same arithmetic): results are
//     bit-identical to the kernel this replaces (tests/test_gpu_ops.py::test_plain256_equals_tile_kernel).
// Served: EPI_BIAS -> fp32, EPI_BIAS_GELU -> f16, EPI_BIAS_LS_RES -> fp32 (in place), EPI_SAM_QKV, EPI_QKV_F16; M >= 2 048, K % 64 == 0.
// Everything else (and the implicit 3 x 3 convolution of the neck) stays on gemm_planes16_kernel.
#include "gemm_core.h"
#include "kernels.h"

namespace {

using gemm_core::EPI_ST;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void* pl_lds_ptr;

constexpr int PL_BM = 256, PL_THREADS = 512;
constexpr float PL_A_SCALE = K_PLANES_ACT_SCALE, PL_W_SCALE = K_PLANES_W_SCALE;
constexpr int pl_stages(int bn) { return bn == 128 ? 3 : 2; }   // 48 KB x 3 or 64 KB x 2
constexpr size_t pl_lds_bytes(int bn) { return size_t(pl_stages(bn)) * (PL_BM + bn) * 128; }
static_assert(size_t(8) * 32 * EPI_ST * sizeof(float) <= pl_lds_bytes(128), "epilogue staging of the eight waves must fit the stages");

#define PL_FENCE() __builtin_amdgcn_sched_barrier(0x76)   // VALU | SALU | VMEM may cross; MFMA and DS may not

__device__ __forceinline__ f32x4 pl_mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// exact-erf GELU on a pair: the arithmetic of gemm_planes.hip:gelu_erf_pair, instruction for instruction (bit-identical)
__device__ __forceinline__ f32x2 pl_gelu_pair(f32x2 x) {
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

// NWN: waves along N (4: 256-column tiles, a wave owns 128 x 64; 2: 128-column tiles, a wave owns 64 x 64)
// X3: the operands are f16x3 PLANES (a row's K-step = [32 hi | 32 lo] halves — the same 128 bytes per row and K-step as 64
// plain f16 columns, so staging, LDS image and fragment addresses are shared): three MFMAs per product (lo.hi, hi.lo, hi.hi,
// gemm_planes.hip's order), planes output.
// CONV (with X3, EPI_CONV; round 4): the implicit 3 x 3 stride-1 convolution of conv.hip — K-step (chunk, tap) reads the A rows
// shifted by (dy * Wp + dx) rows, a scalar offset of the staging instructions; chunk-major K order and W's own K offset as in
// gemm_planes.hip (DESIGN.md findings 19, 27)
// CONV = 2: the stride-2 convolutions (3 x 3 pad 1, or the 1 x 1 shortcut: conv_s2_taps = 9 / 1) WITHOUT conv.hip's gathered tap tensor
// (1.9 GB written and read back per 48-image call): output row R is pixel (b, yo, xo) of the zero-bordered OUTPUT grid, its taps are
// the input rows base(R) + ky * Wpi + kx with base(R) = (b * Hpi + 2 (yo - 1)) * Wpi + 2 (xo - 1) — a per-lane row base (computed
// once per workgroup: one tile each) plus the same scalar tap shift.  K order TAP-major, the gathered GEMM's, so that the two
// routes give the same bits (border rows differ — the gather writes zeros there — and are zeroed by zero_border either way).
template <int EPI, bool OUT_F16, int NWN, bool X3 = false, int CONV = 0>
__global__ __launch_bounds__(PL_THREADS) void gemm_plain256_kernel(const GemmParams g) {
    constexpr int BN = 64 * NWN, SUB = NWN / 2, ROWS = PL_BM + BN, NST = pl_stages(BN);
    constexpr int STAGE = ROWS * 64;   // halves per stage: 128-byte rows, A rows then W rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* const lds = reinterpret_cast<_Float16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    // L2-aware order (round 4): the 32 workgroups an XCD runs at a time are consecutive logical ids; walked row-tile-major
    // inside groups of GROUP_M row tiles they form a 4 x 8 block of tiles — per K-step 12 distinct 32 KB operand panels for 32
    // CUs instead of ~22 (1.6 row tiles x 20 column tiles of a plain row-major walk at N = 5 120): the long-K weights of config
    // 5 (13 MB, four L2s' worth) come through the fabric once per XCD and group.  Worth 1 - 2 % (SAM ViT-H f16 8.16 -> 8.00 ms
    // per image, ViT-L/14 1.495 -> 1.475; groups of 8: 8.07 / 1.463, of 16: 8.30 / 1.498; profiles/r04/config5_tile_order_ab.txt):
    // these GEMMs are not waiting on L2 misses.
    // (f16x3 planes mode — the ViT-S/14 FC1, whose 2.4 MB of weights stay in L2 anyway — keeps the row-major walk: there the
    // grouped order only spreads a row tile's column siblings over more row tiles in flight: fabric traffic 905 -> 1 010 MB per
    // launch at unchanged time, profiles/r04/pmc_summary.txt history)
    constexpr int GROUP_M = X3 ? 1 : 4;
    const int tiles_m = (g.M + PL_BM - 1) / PL_BM, per_group = GROUP_M * tiles_n;
    const int group = tile / per_group, first_m = group * GROUP_M, in_group = tile - group * per_group;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int pid_n = in_group / gm, pid_m = first_m + in_group - pid_n * gm;
    const int m0 = pid_m * PL_BM, n0 = pid_n * BN;
    const unsigned lda4 = unsigned(g.lda) * 4u, ldw4 = unsigned(g.ldw) * 4u;
    const unsigned a_rows = CONV == 1 ? unsigned(g.M) + 2u * unsigned(g.conv_wp) + 2u : CONV == 2 ? unsigned(g.conv_s2_in_rows) : unsigned(g.M);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a_pl), 0, a_rows * lda4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_pl), 0, unsigned(g.N) * ldw4, 0x00020000);
    const int nk = g.K / 32;   // K counts 64-bit column pairs (GemmParams::plain): a K-step = 32 pairs = 64 columns = 128 B per row

    // ---- staging: a wave-instruction moves 8 rows x 128 B; lane (r8 = lane >> 3, position pos = lane & 7) fetches the piece
    // that belongs at its position of the swizzled row: pos ^ r8 (the rows of an instruction start at a multiple of 8).
    // Rows past M / N lie beyond the descriptor's extent and arrive as zeros.
    constexpr int NIW = BN / 64;   // W instructions per wave and K-step (A: 4)
    const int r8 = lane >> 3, piece = (lane & 7) ^ r8;
    unsigned va[4], vw[4];   // (vw[NIW]: an array bound from the local constexpr makes this hipcc's host pass drop the kernel silently)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned arow = unsigned(m0 + 32 * wave + 8 * i + r8);
        if constexpr (CONV == 2) {   // output pixel -> input base row (rows past M: beyond the input too, or harmless garbage that is dropped)
            const unsigned hw = unsigned(g.conv_s2_hpo) * unsigned(g.conv_s2_wpo);
            const unsigned b = arow / hw, rem = arow - b * hw, yo = rem / unsigned(g.conv_s2_wpo), xo = rem - yo * unsigned(g.conv_s2_wpo);
            arow = (b * unsigned(g.conv_s2_hpi) + 2u * yo - 2u) * unsigned(g.conv_wp) + 2u * xo - 2u;   // (border rows wrap around: range-checked zeros)
        }
        va[i] = arow * lda4 + unsigned(piece) * 16u;
    }
#pragma unroll
    for (int i = 0; i < NIW; ++i) vw[i] = unsigned(n0 + (BN / 8) * wave + 8 * i + r8) * ldw4 + unsigned(piece) * 16u;
// (f16x3 planes mode: the activation rows are loaded sc0 nt — FC1 fabric traffic 900 -> 808 MB per launch = 1.07 x algorithmic,
    // FC1 -2 %, and the FC2 launch behind it -3.5 %: profiles/r04/cache_policy_ab.txt)
#define PL_DMA2(stage, aoff, woff)                                                                                                \
    do {                                                                                                                         \
        _Float16* S_ = lds + (stage) * STAGE;                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (pl_lds_ptr)(S_ + (32 * wave + 8 * i_) * 64), 16, va[i_], (aoff), 0, X3 && CONV == 0 ? 3 : 0); \
        _Pragma("unroll") for (int i_ = 0; i_ < NIW; ++i_)                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (pl_lds_ptr)(S_ + (PL_BM + (BN / 8) * wave + 8 * i_) * 64), 16, vw[i_], (woff), 0, 0); \
    } while (0)
    // K-step k: plain / planes GEMM: both operands at k * 128 bytes; CONV: chunk = k / 9, tap = k % 9 = (dy, dx)
    auto stage_kstep = [&](int stage, int k) {
        if constexpr (CONV == 1) {
            const int chunk = k / 9, tap = k - 9 * chunk, dy = tap / 3, dx = tap - 3 * dy;
            PL_DMA2(stage, unsigned(dy * g.conv_wp + dx) * lda4 + unsigned(chunk) * 128u, unsigned(tap * g.conv_cch + chunk) * 128u);
        } else if constexpr (CONV == 2) {   // tap-major: k = tap * cch + chunk; the 1 x 1 shortcut's only tap is the centre
            const int tap = k / g.conv_cch, chunk = k - tap * g.conv_cch;
            const int ky = g.conv_s2_taps == 9 ? tap / 3 : 1, kx = g.conv_s2_taps == 9 ? tap - 3 * (tap / 3) : 1;
            PL_DMA2(stage, unsigned(ky * g.conv_wp + kx) * lda4 + unsigned(chunk) * 128u, unsigned(k) * 128u);
        } else {
            PL_DMA2(stage, k * 128, k * 128);
        }
    };

    // fragment of a 16-row block: row l15, K-chunk q4 of half h of the K-step
    const int a_row = (wm * 64 * SUB + l15) * 64, w_row = (PL_BM + wn * 64 + l15) * 64;
    const int swz0 = 8 * (q4 ^ (l15 & 7)), swz1 = 8 * ((4 + q4) ^ (l15 & 7));

    f32x4 acc[SUB][4][4];
#pragma unroll
    for (int s = 0; s < SUB; ++s)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[s][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // NST stages, NST - 1 K-steps in flight: the narrow tiles (long-K FC2 reads its 168 MB activation from MALL / HBM: a single
    // K-step of ~1 000 cycles does not cover that latency) run three stages, the 256-column tiles two
    stage_kstep(0, 0);
    if (NST == 3 && nk > 1) stage_kstep(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of stage kt have landed (the younger K-step's 4 + NIW instructions may still be in flight) ...
        if (NST == 3 && kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | (4 + NIW));
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        // ... and everyone's; every wave has left the stage of K-step kt - 1, which the next DMA overwrites.  (__syncthreads() is a
        // fence: the compiler drains every LDS-direct load in front of it — s_waitcnt vmcnt(0) in the ISA whatever the line above
        // says — so the third stage of the narrow tiles never has a second K-step in flight.  With a bare s_barrier it has: parity
        // green, LoFTR leg 1 048 / 1 055 against 1 054 / 1 060 pairs/s — no change, the K-step does not wait for its operands:
        // profiles/r04/lds_dma_lab.txt.)
        __syncthreads();
        if (kt + NST - 1 < nk) stage_kstep((kt + NST - 1) % NST, kt + NST - 1);
        const _Float16* S = lds + (kt % NST) * STAGE;
        // A K-step = 2 halves x SUB sub-tiles x 4 row blocks = U units of one activation fragment and four MFMAs (against the
        // half's four W fragments).  A wave that reads a fragment right before its MFMAs waits out the LDS latency every
        // time: the activation fragment of unit u + 3 is requested right behind the MFMAs of unit u (four slots), the W
        // fragments of the second half three units before it starts.  Scheduling fences pin the order of MFMAs and LDS reads.
        if constexpr (X3) {
            // planes: a unit = one 16-row block = its (hi, lo) fragment pair and twelve MFMAs against the K-step's four (hi, lo) W
            // fragment pairs; the pair of unit r + 2 is requested behind the MFMAs of unit r (three slots)
            constexpr int UX = SUB * 4;
            f16x8 wh[4], wl[4], ah[3], al[3];
            auto rd_ax = [&](int r) {
                ah[r % 3] = *reinterpret_cast<const f16x8*>(S + a_row + r * (16 * 64) + swz0);
                al[r % 3] = *reinterpret_cast<const f16x8*>(S + a_row + r * (16 * 64) + swz1);
            };
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                wl[ni] = *reinterpret_cast<const f16x8*>(S + w_row + ni * (16 * 64) + swz1);
                wh[ni] = *reinterpret_cast<const f16x8*>(S + w_row + ni * (16 * 64) + swz0);
            }
            rd_ax(0);
            rd_ax(1);
            PL_FENCE();
#pragma unroll
            for (int r = 0; r < UX; ++r) {
                f32x4(&c)[4] = acc[r >> 2][r & 3];
                // small terms first, term-major over the four accumulators: the order of gemm_planes16_kernel (bit-identical)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) c[ni] = pl_mfma(wl[ni], ah[r % 3], c[ni]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) c[ni] = pl_mfma(wh[ni], al[r % 3], c[ni]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) c[ni] = pl_mfma(wh[ni], ah[r % 3], c[ni]);
                PL_FENCE();
                if (r + 2 < UX) rd_ax(r + 2);
                PL_FENCE();
            }
        } else {
            constexpr int UPH = SUB * 4, U = 2 * UPH;
            f16x8 wf[2][4], af[4];
            auto rd_a = [&](int u) {
                const int h = u / UPH, r = u % UPH;   // r = sub-tile * 4 + row block: rows 16 r of this wave's 64 SUB
                af[u & 3] = *reinterpret_cast<const f16x8*>(S + a_row + r * (16 * 64) + (h ? swz1 : swz0));
            };
            auto rd_w = [&](int h) {
    #pragma unroll
                for (int ni = 0; ni < 4; ++ni) wf[h][ni] = *reinterpret_cast<const f16x8*>(S + w_row + ni * (16 * 64) + (h ? swz1 : swz0));
            };
            rd_w(0);
            rd_a(0);
            rd_a(1);
            rd_a(2);
            PL_FENCE();
    #pragma unroll
            for (int u = 0; u < U; ++u) {
                const int h = u / UPH, r = u % UPH;
                // accumulators hold C^T (A-operand = W fragment, B-operand = activation fragment): gemm_planes.hip
    #pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[r >> 2][r & 3][ni] = pl_mfma(wf[h][ni], af[u & 3], acc[r >> 2][r & 3][ni]);
                PL_FENCE();
                if (u == UPH - 3) rd_w(1);
                if (u + 3 < U) rd_a(u + 3);
                PL_FENCE();
            }
        }
    }

    // ---- epilogue: gemm_planes16_kernel's, one 64 x 64 sub-tile at a time (PLAIN forms: BIAS -> fp32, BIAS_GELU -> f16 row-major
    // value * 8, BIAS_LS_RES -> fp32, SAM_QKV -> f16 operand rows)
    const unsigned c_row_bytes = unsigned(g.ldc) * 4u;   // fp32 rows and f16 rows (ldc in column pairs) alike
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(OUT_F16 ? g.c_pl : static_cast<void*>(g.C), 0,
                                                                        unsigned(g.M) * c_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == EPI_BIAS_LS_RES ? g.res : g.C), 0,
        EPI == EPI_BIAS_LS_RES ? unsigned(g.res_mod > 0 ? g.res_mod : g.M) * unsigned(g.ldres) * 4u : 0u, 0x00020000);
    auto res_row = [&](unsigned row) -> unsigned { return g.res_mod > 0 ? row % unsigned(g.res_mod) : row; };
    const int ec4 = (lane & 15) * 4, elr = lane >> 4;   // row-layout coordinates after the LDS transposition
    constexpr unsigned DROP = 0xFFFFFF00u;
    constexpr float inv = 1.0f / (PL_A_SCALE * PL_W_SCALE);
    const int col = n0 + wn * 64 + ec4;
    const bool col_ok = col < g.N;
    const int colc = col_ok ? col : 0;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, gamma = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) bias = *reinterpret_cast<const f32x4*>(g.bias + colc);
    if constexpr (EPI == EPI_BIAS_LS_RES) {  // res + (v*inv + bias)*gamma = res + v*(inv*gamma) + bias*gamma
        gamma = g.gamma ? *reinterpret_cast<const f32x4*>(g.gamma + colc) : f32x4{1.f, 1.f, 1.f, 1.f};
        bias = bias * gamma;
        gamma = gamma * inv;
    }
    // EPI_SAM_QKV: this wave's 64 columns lie in ONE of q / k / v (dim % 64 == 0); a lane's four columns in one head
    [[maybe_unused]] int sq_head = 0, sq_c = 0, sq_row_h = 0;
    [[maybe_unused]] float sq_scale = 1.0f;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsq = rres, rmap = rres;
    if constexpr (EPI == EPI_SAM_QKV) {
        const int wcol = n0 + wn * 64;
        const int which = __builtin_amdgcn_readfirstlane((wcol < g.N ? wcol : 0) / g.sam_dim);
        const int rem = colc - which * g.sam_dim;
        sq_head = rem / g.sam_hd;
        sq_c = rem - sq_head * g.sam_hd;
        sq_row_h = which == 2 ? g.sam_dv : g.sam_dq;   // plain operand rows hold the hi parts only: Q' [DQ], K' [DQ], V [DV]
        sq_scale = which == 0 ? g.sam_qscale : 1.0f;
        void* dst = which == 0 ? g.sam_q : which == 1 ? g.sam_k : g.sam_v;
        const unsigned bytes = which == 0 ? g.sam_bytes[0] : which == 1 ? g.sam_bytes[1] : g.sam_bytes[2];
        rsq = __builtin_amdgcn_make_buffer_rsrc(dst, 0, bytes, 0x00020000);
        rmap = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(g.sam_rowmap), 0, unsigned(g.M) * 4u, 0x00020000);
    }
    // EPI_CONV: the residual arrives as activation planes (an empty descriptor — no residual — reads zeros); planes rows are wider
    // than N when the channel count is no multiple of 32 (196 -> 224): the padding columns must read as zeros
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rresp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(EPI == EPI_CONV ? g.res_pl : nullptr), 0, EPI == EPI_CONV && g.res_pl ? unsigned(g.M) * unsigned(g.ldres_pl) * 4u : 0u, 0x00020000);
    [[maybe_unused]] const bool col_pad = EPI == EPI_CONV && OUT_F16 && !col_ok && col < g.ldc;
    [[maybe_unused]] float qkv_scale = 1.0f;
    if constexpr (EPI == EPI_QKV_F16) qkv_scale = colc < g.sam_dim ? g.sam_qscale : 1.0f;   // a lane's four columns lie in one of q / k / v
    __syncthreads();   // all waves have finished reading the last stage: the LDS is the epilogue's now
    float* E = smem + wave * 32 * EPI_ST;
    f32x2 amax = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SUB; ++s) {
        const int m_base = m0 + wm * 64 * SUB + s * 64;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    *reinterpret_cast<f32x4*>(&E[(m2 * 16 + l15) * EPI_ST + ni * 16 + 4 * q4]) = acc[s][2 * mh + m2][ni];
            const unsigned row0 = unsigned(m_base + mh * 32 + elr);
            [[maybe_unused]] f32x4 res[8];
            if constexpr (EPI == EPI_BIAS_LS_RES) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 rres, col_ok ? res_row(row0 + 4 * i) * unsigned(g.ldres) * 4u + unsigned(col) * 4u : DROP, 0, 0));
            }
            [[maybe_unused]] unsigned sq_dest[8];
            if constexpr (EPI == EPI_SAM_QKV) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    sq_dest[i] = __builtin_amdgcn_raw_buffer_load_b32(rmap, (row0 + 4 * i) * 4u, 0, 0) + unsigned(sq_head * g.sam_npad);
            }
            if constexpr (EPI == EPI_CONV) {   // shortcut rows: (hi + lo) / 8 (gemm_planes.hip's arithmetic)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned o = (row0 + 4 * i) * unsigned(g.ldres_pl) * 4u + unsigned((col >> 5) * 128 + (col & 31) * 2);
                    const f16x4 rh = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rresp, col_ok ? o : DROP, 0, 0));
                    const f16x4 rl = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rresp, col_ok ? o + 64u : DROP, 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) res[i][e] = (float(rh[e]) + float(rl[e])) * (1.0f / PL_A_SCALE);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&E[(elr + 4 * i) * EPI_ST + ec4]);
                const unsigned off = (row0 + 4 * i) * c_row_bytes;
                [[maybe_unused]] auto store_planes = [&](f32x4 val) {   // planes row: per 32-column chunk [32 hi | 32 lo] halves, value * 8
                    f16x4 hi, lo;
                    pope_amax4x2(amax, val);
                    pope_split4(val * PL_A_SCALE, hi, lo);
                    const unsigned o = col_ok ? off + unsigned((col >> 5) * 128 + (col & 31) * 2) : DROP;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rc, o, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rc, o + 64u, 0, 2);
                };
                if constexpr (EPI == EPI_BIAS) {
                    v = v * inv + bias;
                    if constexpr (X3 && OUT_F16) store_planes(v);
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, col_ok ? off + unsigned(col) * 4u : DROP, 0, 2);
                } else if constexpr (EPI == EPI_SAM_QKV) {
                    v = (v * inv + bias) * sq_scale;
                    pope_amax4x2(amax, v);
                    const unsigned o = col_ok && row0 + 4 * i < unsigned(g.M) ? (sq_dest[i] * unsigned(sq_row_h) + unsigned(sq_c)) * 2u : DROP;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4)), rsq, o, 0, 0);
                } else if constexpr (EPI == EPI_CONV) {
                    v = (v * inv + bias) + res[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(v[e], 0.f) + g.act_slope * __builtin_fminf(v[e], 0.f);
                    if constexpr (OUT_F16) {   // planes (X3), with the zero channel padding
                        f16x4 hi, lo;
                        pope_amax4x2(amax, v);
                        pope_split4(v * PL_A_SCALE, hi, lo);
                        if (col_pad) { hi = f16x4{0, 0, 0, 0}; lo = f16x4{0, 0, 0, 0}; }
                        const unsigned o = col_ok || col_pad ? off + unsigned((col >> 5) * 128 + (col & 31) * 2) : DROP;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rc, o, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rc, o + 64u, 0, 2);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, col_ok ? off + unsigned(col) * 4u : DROP, 0, 2);
                    }
                } else if constexpr (EPI == EPI_QKV_F16) {   // attention operands: f16 row-major, no activation scale (gemm_planes.hip)
                    v = (v * inv + bias) * qkv_scale;
                    pope_amax4x2(amax, v);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4)), rc,
                                                          col_ok ? off + unsigned(col) * 2u : DROP, 0, 0);
                } else if constexpr (EPI == EPI_BIAS_GELU) {
                    v = v * inv + bias;
                    const f32x2 g01 = pl_gelu_pair(f32x2{v[0], v[1]}), g23 = pl_gelu_pair(f32x2{v[2], v[3]});
                    v = f32x4{g01[0], g01[1], g23[0], g23[1]};
                    if constexpr (X3 && OUT_F16) {
                        store_planes(v);
                    } else if constexpr (OUT_F16) {   // f16 row-major, value * 8
                        pope_amax4x2(amax, v);
                        const f16x4 hh = __builtin_convertvector(v * PL_A_SCALE, f16x4);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), rc, col_ok ? off + unsigned(col) * 2u : DROP, 0, 2);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, col_ok ? off + unsigned(col) * 4u : DROP, 0, 2);
                    }
                } else {   // EPI_BIAS_LS_RES: the residual stream is re-read by the next LayerNorm: default cache policy
                    v = res[i] + v * gamma + bias;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, col_ok ? off + unsigned(col) * 4u : DROP, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (EPI == EPI_SAM_QKV || EPI == EPI_QKV_F16)   // attention operands carry no scale
        pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) < POPE_F16_OVERFLOW));
    else if constexpr (OUT_F16)
        pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) * PL_A_SCALE < POPE_F16_OVERFLOW));
}

template <int EPI, bool OUT_F16, int NWN>
int launch_plain(const GemmParams& g, hipStream_t stream) {
    constexpr int BN = 64 * NWN;
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_plain256_kernel<EPI, OUT_F16, NWN>, pl_lds_bytes(BN), lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = ((g.M + PL_BM - 1) / PL_BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_plain256_kernel<EPI, OUT_F16, NWN>), dim3(tiles), dim3(PL_THREADS), pl_lds_bytes(BN), stream, g);
    return pope_check_launch();
}

template <int EPI, bool OUT_F16>
int launch_plain_n(const GemmParams& g, hipStream_t stream) {
    // 256-column tiles wherever the output is at least two of them wide: even at 1.25 rounds of the chip (SAM ViT-H proj / FC2
    // at 4 images: 64 x 5 tiles on 256 CUs) their half-as-many operand bytes per MFMA beat the 128-column tiles' better
    // quantisation (FC2 0.345 -> 0.27 ms, same-box A/B; profiles/r04/config5_f16_gemm_ab.txt)
    return g.N >= 512 ? launch_plain<EPI, OUT_F16, 4>(g, stream) : launch_plain<EPI, OUT_F16, 2>(g, stream);
}

template <int EPI>
int launch_x3(const GemmParams& g, hipStream_t stream) {
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_plain256_kernel<EPI, true, 4, true>, pl_lds_bytes(256), lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = ((g.M + PL_BM - 1) / PL_BM) * ((g.N + 255) / 256);
    hipLaunchKernelGGL((gemm_plain256_kernel<EPI, true, 4, true>), dim3(tiles), dim3(PL_THREADS), pl_lds_bytes(256), stream, g);
    return pope_check_launch();
}

template <bool OUT_PL, int NWN, int CONV = 1>
int launch_conv(const GemmParams& g, hipStream_t stream) {
    constexpr int BN = 64 * NWN;
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_plain256_kernel<EPI_CONV, OUT_PL, NWN, true, CONV>, pl_lds_bytes(BN), lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = ((g.M + PL_BM - 1) / PL_BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_plain256_kernel<EPI_CONV, OUT_PL, NWN, true, CONV>), dim3(tiles), dim3(PL_THREADS), pl_lds_bytes(BN), stream, g);
    return pope_check_launch();
}

}  // namespace

// the stride-2 convolutions of the LoFTR CNN without the gathered tap tensor (planes out, N <= 256): from one round of the CUs
bool pope_wide_conv_s2_supported(const GemmParams& g) {
    if (g.plain || !g.a_pl || !g.w_pl || !g.c_pl || g.epilogue != EPI_CONV || g.conv_cch <= 0 || g.nbatch > 1 || g.res_pl) return false;
    if ((g.conv_s2_taps != 9 && g.conv_s2_taps != 1) || g.K != g.conv_s2_taps * 32 * g.conv_cch || g.lda != 32 * g.conv_cch || g.conv_wp < 4) return false;
    if (g.conv_s2_hpo < 3 || g.conv_s2_wpo < 3 || g.conv_s2_hpi < 4 || g.conv_s2_in_rows <= 0 || (g.ldw & 31) || (g.ldc & 31) || g.N > 256) return false;
    if (size_t(g.conv_s2_in_rows) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N + 256) * g.ldw * 4 >= (size_t(1) << 32) ||
        size_t(g.M + PL_BM) * g.ldc * 4 >= (size_t(1) << 32) - 512)
        return false;
    const int bn = g.N <= 128 ? 128 : 256;
    return size_t((g.M + PL_BM - 1) / PL_BM) * ((g.N + bn - 1) / bn) >= size_t(pope_cu_count());
}

int pope_launch_wide_conv_s2(const GemmParams& g, hipStream_t stream) {
    if (!pope_wide_conv_s2_supported(g)) return POPE_ERR_ARG;
    return g.N <= 128 ? launch_conv<true, 2, 2>(g, stream) : launch_conv<true, 4, 2>(g, stream);
}

// the implicit 3 x 3 convolutions of the LoFTR CNN (conv.hip) at batch size: 256-row tiles, 128 (N <= 128) or 256 columns
bool pope_wide_conv_supported(const GemmParams& g) {
    if (g.plain || !g.a_pl || !g.w_pl || g.epilogue != EPI_CONV || g.conv_cch <= 0 || g.nbatch > 1) return false;
    if (g.K != 9 * 32 * g.conv_cch || g.lda != 32 * g.conv_cch || g.conv_wp < 3 || (g.ldw & 31) || (g.ldc & 31)) return false;
    if ((g.c_pl == nullptr) == (g.C == nullptr) || g.N > 256) return false;
    const int bn = g.N <= 128 ? 128 : 256;
    // from one round of the CUs: with 36 - 72 K-steps per tile the epilogue that nothing runs under is a few per cent (ResNet-FPN call
    // at 48 images 12.1 -> 10.9 - 11.2 ms, at the drivers' 6 images 2.08 -> 1.97; thresholds of 4 / 2 / 1 rounds: profiles/r04/conv_wide_ab.txt)
    return size_t((g.M + PL_BM - 1) / PL_BM) * ((g.N + bn - 1) / bn) >= size_t(pope_cu_count());
}

int pope_launch_wide_conv(const GemmParams& g, hipStream_t stream) {
    if (!pope_wide_conv_supported(g)) return POPE_ERR_ARG;
    if (size_t(g.M + PL_BM + 2 * g.conv_wp + 2) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N + 256) * g.ldw * 4 >= (size_t(1) << 32) ||
        size_t(g.M + PL_BM) * g.ldc * 4 >= (size_t(1) << 32) - 512 || (g.res_pl && size_t(g.M + PL_BM) * g.ldres_pl * 4 >= (size_t(1) << 32) - 512))
        return POPE_ERR_ARG;
    if (g.c_pl) return g.N <= 128 ? launch_conv<true, 2>(g, stream) : launch_conv<true, 4>(g, stream);
    return g.N <= 128 ? launch_conv<false, 2>(g, stream) : launch_conv<false, 4>(g, stream);
}

// f16x3 planes -> planes (QKV, FC1 of the ViT blocks) on the 256 x 256 LDS-direct mainloop: the shapes it serves
bool pope_wide_x3_supported(const GemmParams& g) {
    if (g.plain || !g.a_pl || !g.w_pl || !g.c_pl || g.conv_cch > 0 || g.nbatch > 1) return false;
    if (g.epilogue != EPI_BIAS && g.epilogue != EPI_BIAS_GELU) return false;
    if (g.N < 512 || (g.N & 63) || g.K < 64 || (g.K & 31) || (g.lda & 31) || (g.ldw & 31) || (g.ldc & 31)) return false;
    // 256-row tiles must fill the chip for several rounds: one workgroup per CU, nothing runs under a tile's epilogue
    return size_t((g.M + PL_BM - 1) / PL_BM) * ((g.N + 255) / 256) >= size_t(4) * pope_cu_count();
}

int pope_launch_wide_x3(const GemmParams& g, hipStream_t stream) {
    if (!pope_wide_x3_supported(g)) return POPE_ERR_ARG;
    if (size_t(g.M + PL_BM) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N + 256) * g.ldw * 4 >= (size_t(1) << 32) ||
        size_t(g.M + PL_BM) * g.ldc * 4 >= (size_t(1) << 32) - 512)
        return POPE_ERR_ARG;
    return g.epilogue == EPI_BIAS_GELU ? launch_x3<EPI_BIAS_GELU>(g, stream) : launch_x3<EPI_BIAS>(g, stream);
}

bool pope_plain256_supported(const GemmParams& g) {
    if (!g.plain || !g.a_pl || !g.w_pl || g.conv_cch > 0 || g.nbatch > 1) return false;
    if (g.M < 2048 || g.N < 256 || (g.N & 63) || g.K < 64 || (g.K & 31) || (g.lda & 31) || (g.ldw & 31)) return false;
    const bool out_f16 = g.c_pl != nullptr;
    switch (g.epilogue) {
        case EPI_BIAS: return !out_f16 && g.C;
        case EPI_BIAS_GELU: return out_f16;
        case EPI_BIAS_LS_RES: return !out_f16 && g.C && g.res;
        case EPI_SAM_QKV: return out_f16;
        case EPI_QKV_F16: return out_f16 && g.sam_dim > 0 && !(g.sam_dim & 63);
    }
    return false;
}

int pope_launch_plain256(const GemmParams& g, hipStream_t stream) {
    if (!pope_plain256_supported(g)) return POPE_ERR_ARG;
    if (size_t(g.M + PL_BM) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N + 256) * g.ldw * 4 >= (size_t(1) << 32) ||
        size_t(g.M + PL_BM) * g.ldc * 4 >= (size_t(1) << 32) - 512 || size_t(g.K) * 4 >= (size_t(1) << 31))
        return POPE_ERR_ARG;
    switch (g.epilogue) {
        case EPI_BIAS: return launch_plain_n<EPI_BIAS, false>(g, stream);
        case EPI_BIAS_GELU: return launch_plain_n<EPI_BIAS_GELU, true>(g, stream);
        case EPI_BIAS_LS_RES: return launch_plain_n<EPI_BIAS_LS_RES, false>(g, stream);
        case EPI_SAM_QKV: return launch_plain_n<EPI_SAM_QKV, true>(g, stream);
        case EPI_QKV_F16: return launch_plain_n<EPI_QKV_F16, true>(g, stream);
    }
    return POPE_ERR_ARG;
}
