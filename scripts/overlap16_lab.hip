// Dev harness: do f16-MFMA phases of one wave overlap VALU phases of OTHER waves on the same SIMD?
// (scripts/overlap_lab.hip asked this for the fp32 MFMA, which executes on the VALU lanes and cannot.)
// Each wave alternates NM back-to-back v_mfma_f32_32x32x16_f16 (4 independent accumulators) with NV VALU ops
// (fma / exp2 mix, independent chains).  Reported: wall time per iteration per wave-on-SIMD vs the two pipes' own time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, int MODE, int DEP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ float lds[];
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const int slot = hwid & 0xf;
    if (MODE == 1) {  // distinct static priorities per wave slot
        if ((slot & 3) == 1) __builtin_amdgcn_s_setprio(1);
        else if ((slot & 3) == 2) __builtin_amdgcn_s_setprio(2);
        else if ((slot & 3) == 3) __builtin_amdgcn_s_setprio(3);
    }
    if (MODE == 2) for (int i = 0; i < (slot & 3); ++i) __builtin_amdgcn_s_sleep(40);  // start stagger
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = _Float16(threadIdx.x * 1e-3f + i); y[i] = _Float16(blockIdx.x * 1e-4f - i); }
    float z0 = threadIdx.x * 1e-3f, z1 = z0 + 1.f, z2 = z0 + 2.f, z3 = z0 + 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NM / 4; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, a3, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DEP) { z0 += a0[0] * 1e-20f; }   // the VALU phase starts only after the MFMA phase has finished (like a softmax)
#pragma unroll
        for (int j = 0; j < NV / 8; ++j) {
            z0 = __builtin_fmaf(z0, 1.0001f, 0.5f); z1 = __builtin_fmaf(z1, 1.0001f, 0.5f);
            z2 = __builtin_fmaf(z2, 1.0001f, 0.5f); z3 = __builtin_fmaf(z3, 1.0001f, 0.5f);
            z0 = __builtin_fmaf(z0, 0.9999f, 0.25f); z1 = __builtin_fmaf(z1, 0.9999f, 0.25f);
            z2 = __builtin_amdgcn_exp2f(z2 * 1e-3f); z3 = __builtin_fmaf(z3, 0.9999f, 0.25f);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DEP) { x[0] += _Float16(z0 * 1e-30f); }  // the next MFMA phase depends on the VALU phase
    }
    float s = z0 + z1 + z2 + z3;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 1234.5678f) out[threadIdx.x] = s;
}

// In-wave interleave: after every MFMA, NV/NM VALU ops (independent of the MFMA results).
template <int NM, int NV>
__global__ __launch_bounds__(256) void ki(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = _Float16(threadIdx.x * 1e-3f + i); y[i] = _Float16(blockIdx.x * 1e-4f - i); }
    float z0 = threadIdx.x * 1e-3f, z1 = z0 + 1.f, z2 = z0 + 2.f, z3 = z0 + 3.f;
    constexpr int PER = NV / NM;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            if ((j & 3) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            if ((j & 3) == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
            if ((j & 3) == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
            if ((j & 3) == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, a3, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < PER; ++v) {
                if ((v & 3) == 0) z0 = __builtin_fmaf(z0, 1.0001f, 0.5f);
                if ((v & 3) == 1) z1 = __builtin_fmaf(z1, 0.9999f, 0.25f);
                if ((v & 3) == 2) z2 = (v & 4) ? __builtin_amdgcn_exp2f(z2 * 1e-3f) : __builtin_fmaf(z2, 1.0001f, 0.5f);
                if ((v & 3) == 3) z3 = __builtin_fmaf(z3, 0.9999f, 0.25f);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
        }
    }
    float s = z0 + z1 + z2 + z3;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 1234.5678f) out[threadIdx.x] = s;
}

// In-wave interleave with ONLY transcendental ops (v_exp_f32) or ONLY fma behind each MFMA: cost per op.
template <int PER, int TRANS>
__global__ __launch_bounds__(256) void kt(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = _Float16(threadIdx.x * 1e-3f + i); y[i] = _Float16(blockIdx.x * 1e-4f - i); }
    float z[8];
    for (int i = 0; i < 8; ++i) z[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            if ((j & 3) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            if ((j & 3) == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
            if ((j & 3) == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
            if ((j & 3) == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, a3, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < PER; ++v) {
                if (TRANS) z[v & 7] = __builtin_amdgcn_exp2f(z[v & 7]);
                else z[v & 7] = __builtin_fmaf(z[v & 7], 0.9999f, 0.25f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += z[i];
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 1234.5678f) out[threadIdx.x] = s;
}
template <int PER, int TRANS>
void runt(const char* name, float* out) {
    const int iters = 400;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL((kt<PER, TRANS>), dim3(512), dim3(256), 0, 0, out, iters); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-34s 2 waves/SIMD: %7.1f ns per 24-MFMA iteration per wave (%d ops per MFMA)\n", name, best * 1e6 / iters / 2, PER);
}

template <int NM, int NV>
void runi(const char* name, float* out) {
    const int iters = 400;
    printf("%-34s", name);
    for (int wg : {1, 2, 4}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); hipLaunchKernelGGL((ki<NM, NV>), dim3(256 * wg), dim3(256), 0, 0, out, iters); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("  %dx256 WGs: %7.1f ns/iter/wave", wg, best * 1e6 / iters / wg);
    }
    printf("\n");
}

template <int NM, int NV, int MODE, int DEP>
void run(const char* name, float* out) {
    const int iters = 400;
    printf("%-34s", name);
    for (int wg : {1, 2, 4}) {   // waves per SIMD
        size_t lds = wg == 1 ? 100000 : wg == 2 ? 60000 : 36000;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k<NM, NV, MODE, DEP>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); hipLaunchKernelGGL((k<NM, NV, MODE, DEP>), dim3(256 * wg), dim3(256), lds, 0, out, iters); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("  %dw/SIMD: %7.1f ns/iter/wave", wg, best * 1e6 / iters / wg);
    }
    printf("\n");
}

int main() {
    float* out; hipMalloc(&out, 8192);
    for (int r = 0; r < 2; ++r) {
        run<24, 0, 0, 0>("24 MFMA only", out);
        run<0, 256, 0, 0>("256 VALU only", out);
        run<24, 256, 0, 0>("24 MFMA + 256 VALU indep", out);
        run<24, 256, 0, 1>("24 MFMA -> 256 VALU dependent", out);
        run<24, 256, 1, 1>("  same, setprio by slot", out);
        run<24, 256, 2, 1>("  same, start stagger", out);
        runi<24, 96>("interleaved 24 MFMA + 96 VALU", out);
        runi<24, 192>("interleaved 24 MFMA + 192 VALU", out);
        runi<24, 288>("interleaved 24 MFMA + 288 VALU", out);
        runt<0, 0>("MFMA only", out);
        runt<4, 0>("4 v_fma per MFMA", out);
        runt<8, 0>("8 v_fma per MFMA", out);
        runt<2, 1>("2 v_exp per MFMA", out);
        runt<4, 1>("4 v_exp per MFMA", out);
        runt<8, 1>("8 v_exp per MFMA", out);
    }
    return 0;
}
