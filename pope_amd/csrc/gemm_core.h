// Shared 128x128x32 fp32-MFMA tile mainloop + coalesced epilogue (see gemm_f32.hip for the design).
#pragma once
#include "common.h"
#include <type_traits>

namespace gemm_core {

constexpr int BM = 128, BN = 128, BK = 32, LDS_ST = 36;
constexpr int THREADS = 256;
// ONE LDS stage (A and W tiles of one K-step): 36.9 KB -> three workgroups per CU (VGPR-limited),
// i.e. three waves per SIMD from three independent workgroups: their barrier, LDS-turnaround
// and epilogue bubbles interleave instead of lining up (measured with two lock-stepped
// workgroups per CU and double-buffered LDS: MFMA pipe busy only 60-67 %).
constexpr size_t LDS_BYTES = size_t(BM + BN) * LDS_ST * sizeof(float);
constexpr int EPI_ST = 68;  // epilogue staging row (floats): 17 x 16 B, conflict-free b128 rows
static_assert(size_t(4) * 32 * EPI_ST * sizeof(float) <= LDS_BYTES, "epilogue staging must fit the K-step stage");

// ---- operand loaders -------------------------------------------------------------------------
// A loader serves this thread's four staging pieces of a K-step: piece i = the four K-contiguous
// floats at tile row (tid>>3) + 32*i, columns k0 + (tid&7)*4 .. +3; zeros outside the operand.

// Dense row-major operand [rows, ld] with ld == K-extent % 32 == 0, < 4 GiB: one
// `buffer_load_dwordx4 ... offen` per piece — per-piece byte offset in a loop-invariant VGPR, the
// K-step offset in an SGPR, rows past the end zero-filled by the buffer bounds check.  No
// address arithmetic, no exec-mask branches in the K loop (the plain-pointer loader's guards
// cost ~10 % of the GEMM: lab ablation "noload").
struct BufferLoader {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[4];
    __device__ __forceinline__ BufferLoader(const float* base, int rows, int ld, int row0) {
        // base / rows / ld are kernel arguments (wave-uniform): the descriptor stays in SGPRs
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, unsigned(rows) * unsigned(ld) * 4u, 0x00020000);
        const int srow = threadIdx.x >> 3, scol = (threadIdx.x & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) voff[i] = (unsigned(row0 + srow + 32 * i) * unsigned(ld) + scol) * 4u;
    }
    __device__ __forceinline__ f32x4 load(int i, int k0) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], k0 * 4, 0));
    }
};

// Generic loader: f(tile_row, k) -> four floats (guards, gathers and prescales live in f).
template <class F>
struct FnLoader {
    F f;
    __device__ __forceinline__ f32x4 load(int i, int k0) const {
        return f((threadIdx.x >> 3) + 32 * i, k0 + (threadIdx.x & 7) * 4);
    }
};
template <class F>
__device__ __forceinline__ FnLoader<F> fn_loader(F f) { return FnLoader<F>{f}; }

// acc = (A_tile . W_tile^T)^T over K, i.e. acc[mi][ni] holds C^T: register rows run over n,
// the lane column over m — so a lane owns 4 consecutive output columns per register quad and the
// epilogue can move 16-byte pieces.
// Register prefetch: on entry ra/rw hold K-step 0 of this tile; the global loads of K-step t+1
// are issued right after the LDS stage of K-step t is published and complete under its 64 MFMAs.
// During the LAST K-step the loads of K-step 0 of the workgroup's NEXT tile are issued (nla/nlw,
// if has_next) so that a persistent workgroup streams across tile seams: the epilogue's stores
// drain under the next tile's MFMAs instead of holding the CU slot until HBM has taken them.
// LAB is 0 in the product; scripts/gemm_lab.hip instantiates timing-only ablations:
// bit0 = no global loads after the first K-step, bit1 = no LDS restaging/barriers after the first.
template <int LAB = 0, class LoaderA, class LoaderW>
__device__ __forceinline__ void mainloop_prefetched(const LoaderA& la, const LoaderW& lw, const LoaderA& nla,
                                                    const LoaderW& nlw, bool has_next, int K, float* smem,
                                                    f32x16 (&acc)[2][2], f32x4 (&ra)[4], f32x4 (&rw)[4]) {
    float* As = smem;                // [BM][LDS_ST]
    float* Ws = smem + BM * LDS_ST;  // [BN][LDS_ST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int srow = tid >> 3, scol = (tid & 7) * 4;

    const int nk = (K + BK - 1) / BK;
    const float* a_base = &As[(wm * 64 + r) * LDS_ST + 4 * h];
    const float* w_base = &Ws[(wn * 64 + r) * LDS_ST + 4 * h];
    // One K-step.  The first step of a tile starts its accumulation chains from the MFMA's inline
    // zero C operand instead of zero-filled registers (64 v_mov per tile saved: VALU instructions
    // are paid against the f32 MFMA stream, which runs on the same lanes).
    auto kstep = [&](int kt, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        if (!(LAB & 2) || FIRST) {
            __syncthreads();  // every wave is done reading the previous stage / epilogue staging
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(&As[(srow + 32 * i) * LDS_ST + scol]) = ra[i];
                *reinterpret_cast<f32x4*>(&Ws[(srow + 32 * i) * LDS_ST + scol]) = rw[i];
            }
            __syncthreads();
        }
        if (!(LAB & 1)) {
            if (kt + 1 < nk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ra[i] = la.load(i, (kt + 1) * BK);
                    rw[i] = lw.load(i, (kt + 1) * BK);
                }
            } else if (has_next) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ra[i] = nla.load(i, 0);
                    rw[i] = nlw.load(i, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const f32x4*>(a_base + t * 32 * LDS_ST + 8 * j);
                b[t] = *reinterpret_cast<const f32x4*>(w_base + t * 32 * LDS_ST + 8 * j);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        if (FIRST && j == 0 && s == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[mi][ni] = mfma_32x32x2(b[ni][s], a[mi][s], zero);
                        } else {
                            acc[mi][ni] = mfma_32x32x2(b[ni][s], a[mi][s], acc[mi][ni]);
                        }
                    }
        }
    };
    kstep(0, std::true_type{});
    for (int kt = 1; kt < nk; ++kt) kstep(kt, std::false_type{});
}

// One tile, no cross-tile prefetch.
template <int LAB = 0, class LoaderA, class LoaderW>
__device__ __forceinline__ void mainloop(const LoaderA& la, const LoaderW& lw, int K, float* smem, f32x16 (&acc)[2][2]) {
    f32x4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = la.load(i, 0);
        rw[i] = lw.load(i, 0);
    }
    mainloop_prefetched<LAB>(la, lw, la, lw, false, K, smem, acc, ra, rw);
}

// Coalesced epilogue: each wave transposes its 64x64 sub-tile through a private 32x68-float LDS
// region (two passes of 32 rows; wave-local, no workgroup barrier after the initial one) and
// calls f(row_in_tile, col_in_tile, v) with v = four consecutive output columns; the 16 lanes
// sharing a row cover 256 contiguous bytes, so f's loads/stores are whole 256-B row segments
// and a wave issues 16 wide stores instead of 64 dword stores.
template <class F>
__device__ __forceinline__ void epilogue_rows(const f32x16 (&acc)[2][2], float* smem, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    __syncthreads();  // all waves have finished reading the last K-step stage
    float* E = smem + wave * 32 * EPI_ST;
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
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int lr = (lane >> 4) + 4 * i, c4 = (lane & 15) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&E[lr * EPI_ST + c4]);
            f(wm * 64 + mi * 32 + lr, wn * 64 + c4, v);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace gemm_core
