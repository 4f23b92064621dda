// Shared device/host helpers for the gfx950 (CDNA4, MI355X) kernels of the POPE hot path.
// Wave = 64 lanes everywhere; MFMA = v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, exact
// k-ordered fmaf chain — the reference path is fp32 end to end).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define POPE_OK 0
#define POPE_ERR_ARG (-1)
#define POPE_ERR_LAUNCH (-2)
#define POPE_ERR_WORKSPACE (-3)

// C/D fragment map of every 32x32 MFMA on gfx950: register i of lane l holds
// row (i&3) + 8*(i>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ int mfma32_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Cross-lane exchange over lane distance 16 / 32 in ONE instruction (gfx950 v_permlane16/32_swap_b32): both results
// together hold, in every lane l, the values of lanes l and l ^ 16 (l ^ 32) — in a fixed order per lane, so a
// commutative combine (max, a + b) is deterministic.  (__shfl_xor is a ds_bpermute: address arithmetic + LDS trip.)
__device__ __forceinline__ void pope_xor16_pair(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    // (elements are copied to scalars first: __builtin_bit_cast applied to r[1] directly reads element 0 with this clang)
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void pope_xor32_pair(float v, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    // (elements are copied to scalars first: __builtin_bit_cast applied to r[1] directly reads element 0 with this clang)
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}

// Wave-uniform select that can never become control flow: the compiler turns chains of scalar ?: into branches at will,
// and a branch inside a software-pipelined K-step splits its basic block (the instruction-mix pins then no longer
// reach across it).  c != 0 ? a : b, all three in SGPRs.
__device__ __forceinline__ int pope_uniform_select(int c, int a, int b) {
    int r;   // readfirstlane: the operands are wave-uniform by contract; this pins them to SGPRs for the "s" constraints
    asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, %2, %3"
        : "=s"(r)
        : "s"(__builtin_amdgcn_readfirstlane(c)), "s"(__builtin_amdgcn_readfirstlane(a)), "s"(__builtin_amdgcn_readfirstlane(b))
        : "scc");
    return r;
}

// Blocks b and b+8 share an XCD (round-robin dispatch).  Remap so that each XCD walks a
// contiguous chunk of the logical tile space (neighbouring tiles share operand panels in
// that XCD's private L2).  Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Per-device host state.  The reference keeps the matcher on cuda:1 and DINOv2 on cuda:0 in one process
// (pope_model_api.py:181-184), so everything the launchers cache is keyed by the CURRENT device (the Python
// binding makes the operand's device current around every call).
constexpr int POPE_MAX_DEVICES = 64;
static inline int pope_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= POPE_MAX_DEVICES) d = 0;
    return d;
}

// Number of CUs of the current device (cached per device; 256 on MI355X).  Host-side query, no sync.
static inline int pope_cu_count() {
    static std::atomic<int> cus[POPE_MAX_DEVICES];
    const int dev = pope_current_device();
    int n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// One-time (per kernel AND device) opt-in for more than 64 KB of dynamic LDS.  `done` is a per-kernel bit mask of
// the devices that already have the attribute.
typedef std::atomic<unsigned long long> pope_dev_mask;
template <typename K>
static inline bool pope_opt_in_lds(K kernel, size_t bytes, pope_dev_mask& done) {
    const int dev = pope_current_device();
    if ((done.load(std::memory_order_acquire) >> dev) & 1ull) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)) !=
        hipSuccess)
        return false;
    done.fetch_or(1ull << dev, std::memory_order_release);
    return true;
}

// ---- f16x3 operand split ------------------------------------------------------------------------------------------
// x = hi + lo with hi = f16(x) (RNE) and lo = f16(x - hi): x - hi is exact in fp32, so lo is its correctly rounded f16.
// v_fma_mixlo/mixhi_f16 read the f16 half of `hi` directly and write the f16 result into one half of the destination:
// four instructions produce the four lo halves of a quad (instead of 4 x v_cvt_f32_f16 + 4 x v_sub + 2 x v_cvt_pk) —
// bit-identical results, 6 instead of 12 VALU instructions per quad.
typedef _Float16 pope_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pope_split_lo_pair(float v0, float v1, unsigned hi_pair) {
    // two statements: the first result may share v0's register (v0 is dead after it) — the allocator decides; hi_pair
    // stays live across both, so the half-written destination never aliases it
    unsigned d;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi_pair), "v"(v0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi_pair), "v"(v1));
    return d;
}
__device__ __forceinline__ void pope_split4(f32x4 v, pope_f16x4& hi, pope_f16x4& lo) {
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    hi = __builtin_convertvector(v, pope_f16x4);  // 2 x v_cvt_pk_f16_f32 (RNE)
    const u32x2s hp = __builtin_bit_cast(u32x2s, hi);
    const u32x2s lp = {pope_split_lo_pair(v[0], v[1], hp[0]), pope_split_lo_pair(v[2], v[3], hp[1])};
    lo = __builtin_bit_cast(pope_f16x4, lp);
}

// ---- f16x3 range guard ------------------------------------------------------------------------------------------
// A planes producer converts value * scale to f16; a finite fp32 value whose scaled magnitude reaches 65520 rounds
// to +-inf there (and poisons everything downstream) although the fp32 reference is fine.  Every producer therefore
// tracks the largest scaled magnitude it converts and ORs its POPE_RANGE_* bit into a caller-provided device word;
// the host checks the word at its next synchronisation point and re-runs the work on the fp32 MFMA.
constexpr float POPE_F16_OVERFLOW = 65520.0f;  // smallest magnitude that RNE-rounds to f16 infinity
__device__ __forceinline__ void pope_range_flag(unsigned* flag, unsigned bit, bool bad) {
    if (flag && bad) (void)__hip_atomic_fetch_or(flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float pope_amax4(float m, f32x4 v) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                           __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])), m));
}
// two independent running maxima (m[0] over elements 0-1, m[1] over 2-3): no serial chain through one register
__device__ __forceinline__ void pope_amax4x2(f32x2& m, f32x4 v) {
    m[0] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), m[0]);
    m[1] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])), m[1]);
}
__device__ __forceinline__ float pope_amax2(float m, f32x2 v) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), m);
}

static inline int pope_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? POPE_OK : POPE_ERR_LAUNCH;
}
