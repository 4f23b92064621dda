// LAB KERNEL — not part of libpope_hip.so.  A measured negative result of round 4 (DESIGN.md finding 29; numbers in
// profiles/r04/xstat_*.txt, xstat_in_model_ab.json): correct and bit-identical to the product's 128 x 128 tile kernel, 4-6 %
// faster than it on QKV / FC1 in isolation (0.273 vs 0.290 ms, 0.394 vs 0.412 ms), 7 % / 1 % SLOWER inside the model (0.312 vs
// 0.290, 0.442 vs 0.440 ms; step 95.7 vs 94.8 ms): with one workgroup per CU all 256 CUs fetch their 196 KB activation panels at
// the same three moments of a launch, and nothing computes under those bursts.  Built into a lab copy of the library by
// scripts/xstat_ab.sh (which also patches the dispatch hook into a copy of gemm_f16x3.hip), timed by scripts/xstat_probe.py.
//
// X-stationary f16x3 "planes" NT GEMM for K = 384: the QKV and FC1 projections of the ViT blocks at large M.
//
// Why a second mainloop.  The 128 x 128 kernel of gemm_planes.hip stages BOTH operands of every tile through LDS (392 KB
// from L2 per 128 x 128 x 384 tile, 196 KB of ds_write) and every LDS instruction costs the matrix pipe its full LDS-array
// time on this chip: ablation builds of this file (scripts/xstat_ab.sh, profiles/r04/xstat_ablation.txt) put one
// ds_read_b128 per wave and K-step at 7.2 us per QKV launch whether 0, 8, 10 or 16 of them are issued, next to MFMAs that
// alone take 151 us — LDS traffic and MFMAs add up, they do not overlap.  So the lever is LDS instructions per MFMA.
// Here a workgroup (8 waves) owns 128 ROWS of the activation for ALL N output columns:
//   * the hi plane of its 128 x 384 activation panel lives in REGISTERS for the whole tile: wave (rg, ch) keeps rows
//     32 rg .. 32 rg + 31 as MFMA B-operand fragments (2 row blocks x 12 K-steps x 4 VGPRs = 96 registers), loaded once
//     from HBM; the lo plane (used by one of the three partial products) sits in LDS for the whole tile (96 KB, XOR-swizzled
//     64-byte rows) and costs two fragment reads per K-step;
//   * only W streams through LDS: 16 KB per K-step, three stages, one barrier per K-step; W is a few MB that every CU
//     reads in the same order (L2 hits); the activation is read from HBM exactly once (the 128 x 128 kernel re-reads each
//     row N / 128 = 9-12 times through L2 / MALL);
//   * a wave computes 32 rows x 64 columns of the current 128-column chunk: 10 fragment reads feed 24 MFMAs per K-step
//     (0.42 per MFMA; the 64 x 64 wave tile: 0.33, but with 2x the staging stores and L2 -> LDS bytes per MFMA).
// The arithmetic is the 128 x 128 kernel's, accumulator by accumulator (W fragment = MFMA A operand, activation
// fragment = B operand, per K-step lo.hi, hi.lo, hi.hi, K-steps ascending) and so is the epilogue's: results are
// bit-identical to gemm_planes16_kernel, which still serves every other shape (and small M, where 128-row tiles alone
// cannot fill the chip).  scripts/xstat_probe.py compares the two bit for bit.
//
// Epilogue.  The C^T accumulator block gives a lane four consecutive columns of one row; two v_permlane16_swap per
// register pair turn two neighbouring blocks into 16-byte pieces of the [32 hi | 32 lo] planes rows (a row's four lanes
// write a whole 64-byte half line per store), so no LDS transposition is needed.  With OVERLAP the epilogue of chunk c
// runs in slices between the MFMAs of chunk c + 1 (its accumulators are kept in a second register set): the matrix pipe
// does not idle under bias / GELU / split arithmetic, which in the 128 x 128 kernel the CU's second workgroup covers.
#include "../pope_amd/csrc/gemm_core.h"
#include "../pope_amd/csrc/kernels.h"

#ifndef XS_LAB
#define XS_LAB 0   // lab builds only (timing ablations, wrong results): 1 no W fragment reads, 2 no W staging, 4 no MFMAs, 8 no barriers, 16 no stores
#endif
#ifndef XS_STORE_AUX
#define XS_STORE_AUX 2   // cache policy of the planes stores: 2 = nt (write-once output), 0 = default
#endif

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void* xs_lds_ptr;

constexpr int XS_BM = 128, XS_BN = 128, XS_THREADS = 512;
constexpr int XS_STAGE = XS_BN * 64;         // halves per W stage: 128 rows x 128 B, 16-byte piece p of row r at position p ^ (r & 7)
constexpr int XS_NSTAGE = 3;
constexpr float XS_A_SCALE = K_PLANES_ACT_SCALE, XS_W_SCALE = K_PLANES_W_SCALE;
constexpr size_t xs_lds_bytes(int nk) { return size_t(nk) * XS_BM * 64 + size_t(XS_NSTAGE) * XS_STAGE * sizeof(_Float16); }   // 147 456 at K = 384

__device__ __forceinline__ f32x4 xs_mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// exact-erf GELU on a pair: the arithmetic of gemm_planes.hip:gelu_erf_pair, instruction for instruction (bit-identical)
__device__ __forceinline__ f32x2 xs_gelu_pair(f32x2 x) {
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

// lanes l and l ^ 16 exchange: afterwards (a, b) of the even 16-lane row hold both rows' `a`, of the odd row both rows' `b`
__device__ __forceinline__ void xs_swap16(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = r0;
    b = r1;
}

template <int EPI, int NK, bool OVERLAP>
__global__ __launch_bounds__(XS_THREADS) void gemm_xstat_kernel(const GemmParams g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS: the lo plane of the activation panel [NK][128 rows][64 B] (16-byte piece q of row r at position q ^ ((r >> 1) & 3):
    // conflict-free 16-row fragment reads), then three W stages
    _Float16* const xl_lds = reinterpret_cast<_Float16*>(smem);
    _Float16* const w_lds = xl_lds + NK * XS_BM * 32;
    static_assert(NK % 6 == 0, "stage (kt % 3) and register set (kt & 1) are compile-time per K-step");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave >> 1, ch = wave & 1;   // rows 32 rg .. + 31 of the tile; columns 64 ch .. + 63 of every 128-column chunk
    const int l15 = lane & 15, q4 = lane >> 4;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * XS_BM;
    const int NC = g.N / XS_BN;   // launcher: N % 128 == 0
    const unsigned lda4 = unsigned(g.lda) * 4u, ldw4 = unsigned(g.ldw) * 4u, ldc4 = unsigned(g.ldc) * 4u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a_pl), 0, unsigned(g.M) * lda4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_pl), 0, unsigned(g.N) * ldw4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(g.c_pl, 0, unsigned(g.M) * ldc4, 0x00020000);
    constexpr unsigned DROP = 0xFFFFFF00u;   // beyond every buffer extent: the access is discarded

#if XS_LAB & 128   // lab: de-phase the workgroups of the first round (chunk ends = store bursts of all CUs otherwise coincide)
    for (int i = (blockIdx.x < 256 ? (blockIdx.x >> 3) & 7 : 0); i > 0; --i) __builtin_amdgcn_s_sleep(39);
#endif
    // ---- the activation panel.  lo plane -> LDS straight from memory (buffer_load ... lds: a wave-instruction fills 1 KB =
    // 16 rows x 64 B; the swizzle is applied on the source side); hi plane -> registers.  Rows >= M read zeros.
    {
        const int row = tid >> 2, pos = tid & 3;
        const unsigned voff = unsigned(m0 + row) * lda4 + 64u + unsigned(pos ^ ((row >> 1) & 3)) * 16u;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (xs_lds_ptr)(xl_lds + kt * (XS_BM * 32) + wave * 512), 16, voff, kt * 128, 0, 0);
    }
    f16x8 xh[2][NK];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const unsigned xoff = unsigned(m0 + 32 * rg + 16 * rb + l15) * lda4 + unsigned(q4) * 16u;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
            xh[rb][kt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, xoff, kt * 128, 0));
    }
    // lo fragment of row block rb at K-step kt
    const _Float16* const xl_base = xl_lds + (32 * rg + l15) * 32 + 8 * (q4 ^ ((l15 >> 1) & 3));
    auto read_xl = [&](int rb, int kt) -> f16x8 { return *reinterpret_cast<const f16x8*>(xl_base + kt * (XS_BM * 32) + rb * (16 * 32)); };

    // ---- W stream: item = (chunk nc, K-step kt) = 128 W rows x 128 B.  Thread -> rows srow, srow + 64, the piece that lands at
    // position pc of the swizzled row.  Past the end of the stream the last item is loaded again and never consumed.
    const int srow = tid >> 3, pc = tid & 7;
    const unsigned wv0 = unsigned(srow) * ldw4 + unsigned(pc ^ (srow & 7)) * 16u, wv1 = wv0 + 64u * ldw4;
    int ld_soff = 0, ld_kt = 0, ld_left = NC * NK - 1;   // items after the one ld_soff points at
    const int chunk_step = XS_BN * int(ldw4) - (NK - 1) * 128;
    auto load_next = [&](u32x4 (&st)[2]) {
        st[0] = __builtin_amdgcn_raw_buffer_load_b128(rw, wv0, ld_soff, 0);
        st[1] = __builtin_amdgcn_raw_buffer_load_b128(rw, wv1, ld_soff, 0);
        const int more = ld_left > 0;   // scalar selects: a K-step stays one basic block
        const int wrap = ++ld_kt == NK;
        ld_soff += pope_uniform_select(more, pope_uniform_select(wrap, chunk_step, 128), 0);
        ld_kt = pope_uniform_select(wrap, 0, ld_kt);
        ld_left -= more;
    };
    _Float16* const wr_base = w_lds + srow * 64 + 8 * pc;
    auto write_stage = [&](int s3, const u32x4 (&st)[2]) {
        *reinterpret_cast<u32x4*>(wr_base + s3 * XS_STAGE) = st[0];
        *reinterpret_cast<u32x4*>(wr_base + s3 * XS_STAGE + 64 * 64) = st[1];
    };
    // fragments of column block j (of this wave's four): W rows 64 ch + 16 j + l15 of the stage, K-chunk q4 of the hi / lo plane
    const _Float16* const fr_hi = w_lds + (64 * ch + l15) * 64 + 8 * (q4 ^ (l15 & 7));
    const _Float16* const fr_lo = w_lds + (64 * ch + l15) * 64 + 8 * ((4 + q4) ^ (l15 & 7));

    // ---- epilogue pieces (the arithmetic of gemm_planes16_kernel's epilogue)
    constexpr float inv = 1.0f / (XS_A_SCALE * XS_W_SCALE);
    f32x2 amax = {0.f, 0.f};
    // after the lane exchange: q4 = 0 holds columns 0-7 of the 32-column chunk, 1: 16-23, 2: 8-15, 3: 24-31
    const unsigned piece_off = unsigned((q4 & 1) * 32 + (q4 >> 1) * 16);
    // no bias = an empty descriptor: every load returns zeros (no branch in the K-steps that carry the loads)
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.bias), 0, g.bias ? unsigned(g.N) * 4u : 0u, 0x00020000);
    auto bias_of = [&](int nc, int j) -> f32x4 {   // the lane's four columns of block j
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_, unsigned(ch * 256 + j * 64 + 16 * q4), nc * (XS_BN * 4), 0));
    };
    auto act = [&](f32x4 v, f32x4 bias) -> f32x4 {
        v = v * inv + bias;
        if constexpr (EPI == EPI_BIAS_GELU) {
            const f32x2 g01 = xs_gelu_pair(f32x2{v[0], v[1]}), g23 = xs_gelu_pair(f32x2{v[2], v[3]});
            v = f32x4{g01[0], g01[1], g23[0], g23[1]};
        }
        return v;
    };
#ifndef XS_SPREAD
#define XS_SPREAD 1
#endif
#if XS_SPREAD
    u32x4 sq[8];
    unsigned sq_off[4] = {DROP, DROP, DROP, DROP};
#pragma unroll
    for (int i = 0; i < 8; ++i) sq[i] = u32x4{0u, 0u, 0u, 0u};
#endif
    // planes chunk c (32 columns = blocks 2 c, 2 c + 1 of this wave's 64) of row block rb, N-chunk nc: split, exchange, two
    // 16-byte stores
    auto store_chunk = [&](int nc, int rb, int c, f32x4 va, f32x4 vb, bool live) {
        pope_amax4x2(amax, va);
        pope_amax4x2(amax, vb);
        f16x4 ha, la, hb, lb;
        pope_split4(va * XS_A_SCALE, ha, la);
        pope_split4(vb * XS_A_SCALE, hb, lb);
        const u32x2 pha = __builtin_bit_cast(u32x2, ha), phb = __builtin_bit_cast(u32x2, hb);
        const u32x2 pla = __builtin_bit_cast(u32x2, la), plb = __builtin_bit_cast(u32x2, lb);
        unsigned h0 = pha[0], h1 = pha[1], h2 = phb[0], h3 = phb[1], l0 = pla[0], l1 = pla[1], l2 = plb[0], l3 = plb[1];
        xs_swap16(h0, h2);
        xs_swap16(h1, h3);
        xs_swap16(l0, l2);
        xs_swap16(l1, l3);
        const int row = m0 + 32 * rg + 16 * rb + l15;
        unsigned o = live && row < g.M && !(XS_LAB & 16) ? unsigned(row) * ldc4 + unsigned((nc * 4 + 2 * ch + c) * 128) + piece_off : DROP;
        if ((XS_LAB & 256) && o != DROP) o &= 0x1FFF80u;   // lab: every store lands in one 2 MB window (no HBM write stream)
#if XS_SPREAD   // lab: the chunk's eight stores are queued and issued one per K-step of the next chunk
        sq[(rb * 2 + c) * 2] = u32x4{h0, h1, h2, h3};
        sq[(rb * 2 + c) * 2 + 1] = u32x4{l0, l1, l2, l3};
        sq_off[rb * 2 + c] = o;
#else
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{h0, h1, h2, h3}, rc, o, 0, XS_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{l0, l1, l2, l3}, rc, o, 64, XS_STORE_AUX);
#endif
    };
#if XS_SPREAD
    auto issue_queued = [&](int i) { __builtin_amdgcn_raw_buffer_store_b128(sq[i], rc, sq_off[i >> 1], (i & 1) * 64, XS_STORE_AUX); };
#endif

    // ---- fragment pipeline.  A wave that reads fragments right before their MFMAs stalls for the LDS latency every time.
    // Units are numbered across K-steps and chunks: unit U = (K-step kt, column block j) = 6 MFMAs (two row blocks x three
    // partial products) fed by one (hi, lo) pair of W fragments in slot U % 3; the reads of unit U + 3 are issued right
    // behind the MFMAs of unit U.  The K-step's barrier sits behind unit 1: everything read after it belongs to the NEXT
    // item, so the pipeline runs through K-step and chunk seams, and the stage of the CURRENT item is free from there on
    // (all its reads were issued before the barrier and the barrier's lgkmcnt(0) has retired them): item + 3 is written
    // into it right away and is published two barriers later.  The lo fragments of the activation for K-step kt + 1 are
    // read during K-step kt.  Scheduling fences pin the order of MFMAs and LDS accesses and let VALU / SALU / VMEM
    // instructions pass (the epilogue slices spread between the MFMAs by themselves).
#ifndef XS_FENCE_MASK
#define XS_FENCE_MASK 0x76   // VALU | SALU | VMEM may cross; MFMA and DS may not
#endif
#define XS_FENCE() __builtin_amdgcn_sched_barrier(XS_FENCE_MASK)
    f16x8 fh[3], fl[3], xlf[2][2];
    bool lab_first = true;
    auto read_unit = [&](int slot, int s3, int j) {
        if ((XS_LAB & 1) && !lab_first) return;
        fh[slot] = *reinterpret_cast<const f16x8*>(fr_hi + s3 * XS_STAGE + j * (16 * 64));
        fl[slot] = *reinterpret_cast<const f16x8*>(fr_lo + s3 * XS_STAGE + j * (16 * 64));
    };

    // ---- prologue: items 0, 1, 2 -> stages 0, 1, 2; items 3 .. 3 + XS_NSETS - 1 in flight in the register sets; units 0-2 and
    // the lo fragments of K-step 0
#ifndef XS_NSETS
#define XS_NSETS 2
#endif
    static_assert(NK % XS_NSETS == 0, "register set (kt % XS_NSETS) is compile-time per K-step");
    u32x4 rs[XS_NSETS][2];
    load_next(rs[0]);
    load_next(rs[1]);
    write_stage(0, rs[0]);
    write_stage(1, rs[1]);
    load_next(rs[0]);
    write_stage(2, rs[0]);
#pragma unroll
    for (int i = 0; i < XS_NSETS; ++i) load_next(rs[i]);
    // (vmcnt retires in issue order: the wait the third stage write needed has also landed the lo plane's LDS-DMA pieces)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) read_unit(j, 0, j);
    xlf[0][0] = read_xl(0, 0);
    xlf[0][1] = read_xl(1, 0);
    lab_first = false;

    f32x4 accA[2][4], accB[2][4];
    // One N-chunk: 12 K-steps into `acc`; OVERLAP: the epilogue of the previous chunk (`pacc`, chunk nc - 1; dead when
    // !plive) in slices behind the K-steps' MFMAs.
    auto chunk = [&](int nc, f32x4 (&acc)[2][4], f32x4 (&pacc)[2][4], bool plive) {
        [[maybe_unused]] f32x4 pb[2];
        const int pnc = nc > 0 ? nc - 1 : 0;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = (kt * 4 + j) % 3;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f32x4 c0 = kt == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[rb][j];
                    if constexpr (XS_LAB & 4) {
                        asm volatile("" ::"v"(fl[slot]), "v"(fh[slot]), "v"(xlf[kt & 1][rb]));
                        acc[rb][j] = c0;
                    } else {
                        f32x4 c = xs_mfma(fl[slot], xh[rb][kt], c0);
                        c = xs_mfma(fh[slot], xlf[kt & 1][rb], c);
                        acc[rb][j] = xs_mfma(fh[slot], xh[rb][kt], c);
                    }
                }
                XS_FENCE();
                if (j == 1) {
                    if (!(XS_LAB & 8)) __syncthreads();   // item kt + 1 is complete; everyone has left item kt's stage
                    if (!(XS_LAB & 2)) {                   // item kt + 3 -> stage kt % 3
                        write_stage(kt % 3, rs[kt % XS_NSETS]);
                        load_next(rs[kt % XS_NSETS]);
                    }
                    XS_FENCE();
                }
                const int nu = j + 3, nkt = kt + (nu >> 2);   // nkt == NK: K-step 0 of the next chunk (NK % 3 == 0)
                read_unit(slot, nkt % 3, nu & 3);
                if (j == 0) {   // lo fragments of the activation for the next K-step (next chunk: K-step 0 again)
                    xlf[(kt + 1) & 1][0] = read_xl(0, (kt + 1) % NK);
                    xlf[(kt + 1) & 1][1] = read_xl(1, (kt + 1) % NK);
                }
                XS_FENCE();
#if XS_SPREAD
                if (j == 3 && kt < 8) issue_queued(kt);
#endif
                if constexpr (OVERLAP) {
                    // K-steps 3 s .. 3 s + 2 carry slice s = (row block, planes chunk) of the previous N-chunk: bias loads,
                    // activation, store
                    const int sl = kt / 3, ph = kt % 3, prb = sl >> 1, pc2 = sl & 1;
                    if (ph == 0 && j == 0) {
                        pb[0] = bias_of(pnc, 2 * pc2);
                        pb[1] = bias_of(pnc, 2 * pc2 + 1);
                    } else if (ph == 1 && j == 0) {
                        pacc[prb][2 * pc2] = act(pacc[prb][2 * pc2], pb[0]);
                    } else if (ph == 1 && j == 2) {
                        pacc[prb][2 * pc2 + 1] = act(pacc[prb][2 * pc2 + 1], pb[1]);
                    } else if (ph == 2 && j == 0) {
                        store_chunk(nc - 1, prb, pc2, pacc[prb][2 * pc2], pacc[prb][2 * pc2 + 1], plive);
                    }
                }
            }
        }
    };
    auto drain = [&](int nc, f32x4 (&acc)[2][4]) {
        f32x4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bias_of(nc, j);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                store_chunk(nc, rb, c, act(acc[rb][2 * c], b[2 * c]), act(acc[rb][2 * c + 1], b[2 * c + 1]), true);
    };

    if constexpr (OVERLAP) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) accB[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int nc = 0;
        for (; nc + 1 < NC; nc += 2) {
            chunk(nc, accA, accB, nc > 0);
            chunk(nc + 1, accB, accA, true);
        }
        if (nc < NC) {   // odd chunk count: one more into A (carrying B's epilogue), then drain A
            chunk(nc, accA, accB, nc > 0);
            drain(nc, accA);
        } else {
            drain(NC - 1, accB);
        }
    } else {
        for (int nc = 0; nc < NC; ++nc) {
            chunk(nc, accA, accB, false);
            drain(nc, accA);
        }
    }
#if XS_SPREAD
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_queued(i);
#endif
    pope_range_flag(g.range_flag, g.range_bit, !(__builtin_fmaxf(amax[0], amax[1]) * XS_A_SCALE < POPE_F16_OVERFLOW));
}

template <int EPI>
int launch_xstat(const GemmParams& g, hipStream_t stream) {
    constexpr int NK = 12;
#ifdef POPE_XSTAT_OVERLAP   // lab builds: the epilogue of chunk c sliced between the MFMAs of chunk c + 1 (slower: see DESIGN.md)
    constexpr bool OVL = true;
#else
    constexpr bool OVL = false;
#endif
    static pope_dev_mask lds_ok{0};  // per kernel instantiation, per device
    if (!pope_opt_in_lds(gemm_xstat_kernel<EPI, NK, OVL>, xs_lds_bytes(NK), lds_ok)) return POPE_ERR_LAUNCH;
    const int tiles = (g.M + XS_BM - 1) / XS_BM;
    hipLaunchKernelGGL((gemm_xstat_kernel<EPI, NK, OVL>), dim3(tiles), dim3(XS_THREADS), xs_lds_bytes(NK), stream, g);
    return pope_check_launch();
}

}  // namespace

// Shapes this mainloop takes; everything else (and every small M) stays on gemm_planes16_kernel, with the same bits.
bool pope_xstat_supported(const GemmParams& g) {
    if (g.plain || !g.c_pl || !g.a_pl || !g.w_pl) return false;
    if (g.epilogue != EPI_BIAS && g.epilogue != EPI_BIAS_GELU) return false;
    if (g.K != 384 || (g.N % XS_BN) || g.N < 2 * XS_BN || (g.lda & 31) || (g.ldw & 31) || (g.ldc & 31)) return false;
    // 128-row tiles alone must fill the chip: below two tiles per CU the 128 x 128 kernel's N / 128 tiles per row block do better
    return (g.M + XS_BM - 1) / XS_BM >= 2 * pope_cu_count();
}

int pope_launch_xstat(const GemmParams& g, hipStream_t stream) {
    if (!pope_xstat_supported(g)) return POPE_ERR_ARG;
    if (size_t(g.M + XS_BM) * g.lda * 4 >= (size_t(1) << 32) || size_t(g.N + 2 * XS_BN) * g.ldw * 4 >= (size_t(1) << 31) ||
        size_t(g.M + XS_BM) * g.ldc * 4 >= (size_t(1) << 32) - 512)
        return POPE_ERR_ARG;
    return g.epilogue == EPI_BIAS_GELU ? launch_xstat<EPI_BIAS_GELU>(g, stream) : launch_xstat<EPI_BIAS>(g, stream);
}
