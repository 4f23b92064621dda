// Dev harness: do MFMA phases of one wave overlap VALU / barrier phases of other waves on the SIMD?
#include "../pope_amd/csrc/common.h"
#include <cstdio>

template <int VALU, int BAR, int MODE = 0>
__global__ __launch_bounds__(256, 3) void k(float* out, int iters) {
    extern __shared__ float lds[];
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID, all 32 bits
    const int slot = hwid & 0xf;
    if (MODE == 1) {  // static priority by SIMD wave slot
        if ((slot % 3) == 0) __builtin_amdgcn_s_setprio(3);
        else if ((slot % 3) == 1) __builtin_amdgcn_s_setprio(2);
    } else if (MODE == 2) {  // start stagger by wave slot
        for (int i = 0; i < (slot % 3); ++i) __builtin_amdgcn_s_sleep(24);
    } else if (MODE == 3) {
        if (blockIdx.x < 8 && (threadIdx.x & 63) == 0) out[1024 + blockIdx.x * 4 + (threadIdx.x >> 6)] = __uint_as_float(hwid);
    }
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f, z = x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a0 = mfma_32x32x2(x, y, a0); a1 = mfma_32x32x2(y, x, a1);
            a2 = mfma_32x32x2(x, x, a2); a3 = mfma_32x32x2(y, y, a3);
        }
        if (VALU) {
#pragma unroll
            for (int j = 0; j < VALU; ++j) z = __builtin_fmaf(z, 1.0001f, a0[j & 15] * 1e-9f);  // depends on the MFMA result
        }
        if (BAR) __syncthreads();
    }
    float s = z; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 1234.5678f) out[threadIdx.x] = s;
}

template <int VALU, int BAR, int MODE = 0>
void run(const char* name, float* out) {
    const int iters = 200;
    for (int wg = 1; wg <= 3; ++wg) {
        size_t lds = wg == 1 ? 100000 : wg == 2 ? 60000 : 40000;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k<VALU, BAR, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); hipLaunchKernelGGL((k<VALU, BAR, MODE>), dim3(256 * wg * 2), dim3(256), lds, 0, out, iters); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double mf = double(256 * wg * 2) * 4 * iters * 64;  // MFMA wave-instructions
        printf("%-28s %d wg/CU: %.3f ms  MFMA pipe util %.1f%% (at 2.4 GHz)\n", name, wg, best,
               100.0 * mf * 64 / (1024.0 * best * 1e-3 * 2.4e9));
    }
}

int main() {
    float* out; hipMalloc(&out, 8192);
    run<128, 1, 3>("hwid dump", out);
    unsigned h[32]; hipMemcpy(h, out + 1024, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) printf("blk %d wave %d: hwid %08x wave_id %u simd %u cu %u se %u\n", i / 4, i % 4, h[i], h[i] & 15, (h[i] >> 4) & 3, (h[i] >> 8) & 15, (h[i] >> 13) & 7);
    for (int r = 0; r < 2; ++r) {
        run<256, 0, 0>("mfma+256 VALU  base", out);
        run<256, 0, 1>("mfma+256 VALU  setprio", out);
        run<256, 0, 2>("mfma+256 VALU  stagger", out);
        run<256, 1, 0>("mfma+256 VALU+bar base", out);
        run<256, 1, 1>("mfma+256 VALU+bar setprio", out);
        run<256, 1, 2>("mfma+256 VALU+bar stagger", out);
    }
}
