// Dev lab (not shipped): what does the CU's memory -> LDS path sustain?  The GEMM mainloops' staging pattern alone (no MFMA):
// a 512-thread workgroup per CU moves K-steps of ROWS x 128 B into LDS stages, a wave instruction = 8 rows x 128 B, with
//   MODE 0: buffer_load ... lds (LDS-direct, what the kernels use)      MODE 1: buffer_load into VGPRs + ds_write_b128
// DEPTH = K-steps in flight (1 or 2), src = a panel that stays in L2 (re-read every K-step) or a stream through HBM.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lds_dma_lab.hip -o scripts/_lab/lds_dma_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH, int ROWS>
__global__ __launch_bounds__(512) void stage_kernel(const char* src, size_t panel_bytes, size_t wg_stride, int ksteps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int STAGE = ROWS * 128, IPW = ROWS / 64;   // instructions per wave and K-step (8 rows each, 8 waves)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xFFFFFFFFu, 0x00020000);
    const int r8 = lane >> 3, piece = (lane & 7) ^ r8;
    unsigned acc = 0;
    auto issue = [&](int stage, int k) {
        const unsigned koff = unsigned((size_t(k) * STAGE) % panel_bytes);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const unsigned row = unsigned(wave * IPW * 8 + i * 8 + r8);
            const unsigned voff = koff + row * 128u + unsigned(piece) * 16u;
            if constexpr (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + stage * STAGE + (wave * IPW + i) * 1024), 16, voff, 0, 0, 0);
            } else {
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
                *reinterpret_cast<u32x4*>(lds + stage * STAGE + (wave * IPW + i) * 1024 + lane * 16) = v;
            }
        }
    };
    for (int d = 0; d < DEPTH; ++d) issue(d, d);
    for (int k = 0; k < ksteps; ++k) {
        if (MODE == 0) {
            if (DEPTH == 2 && k + 1 < ksteps) __builtin_amdgcn_s_waitcnt(0x0f70 | IPW);
            else __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        __syncthreads();
        acc += *reinterpret_cast<const unsigned*>(lds + (k % (DEPTH + 1)) * 0 + ((k % DEPTH) * STAGE) + ((tid * 68) % STAGE));   // one consumer read per thread
        __syncthreads();
        if (k + DEPTH < ksteps) issue((k + DEPTH) % DEPTH == 0 ? 0 : (k + DEPTH) % DEPTH, k + DEPTH);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int DEPTH, int ROWS>
void run(const char* name, const char* src, size_t panel, size_t stride, int wgs) {
    const int ksteps = 2000;
    const size_t lds = size_t(DEPTH) * ROWS * 128;
    hipFuncSetAttribute((const void*)stage_kernel<MODE, DEPTH, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    unsigned* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stage_kernel<MODE, DEPTH, ROWS>), dim3(wgs), dim3(512), lds, 0, src, panel, stride, 50, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stage_kernel<MODE, DEPTH, ROWS>), dim3(wgs), dim3(512), lds, 0, src, panel, stride, ksteps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = double(ksteps) * ROWS * 128;
    printf("%-44s %4d WGs  %7.1f GB/s per CU   %6.2f us per K-step of %d KB   (%.2f TB/s chip)\n", name, wgs, bytes / ms / 1e6, ms * 1e3 / ksteps,
           ROWS * 128 / 1024, bytes * wgs / ms / 1e9);
    hipFree(sink);
}

int main() {
    char* buf; const size_t total = size_t(3) << 30;
    hipMalloc(&buf, total); hipMemset(buf, 1, total);
    // L2-resident: every WG re-reads the same 1 MB panel (all WGs the same one: stride 0)
    run<0, 1, 512>("lds-direct, 1 K-step in flight, L2 panel", buf, 1 << 20, 0, 256);
    run<0, 2, 512>("lds-direct, 2 K-steps in flight, L2 panel", buf, 1 << 20, 0, 256);
    run<0, 2, 256>("lds-direct, 2 x 32 KB in flight, L2 panel", buf, 1 << 20, 0, 256);
    run<1, 1, 512>("vgpr + ds_write, 1 K-step, L2 panel", buf, 1 << 20, 0, 256);
    run<0, 2, 512>("lds-direct, 2 in flight, L2 panel, 1 WG", buf, 1 << 20, 0, 1);
    run<0, 2, 512>("lds-direct, 2 in flight, L2 panel, 32 WGs", buf, 1 << 20, 0, 32);
    // per-WG panels of 8 MB (256 x 8 MB = 2 GB: streams through HBM / MALL)
    run<0, 1, 512>("lds-direct, 1 in flight, 8 MB per WG (HBM)", buf, 8 << 20, 8 << 20, 256);
    run<0, 2, 512>("lds-direct, 2 in flight, 8 MB per WG (HBM)", buf, 8 << 20, 8 << 20, 256);
    run<1, 1, 512>("vgpr + ds_write, 1 K-step, 8 MB per WG (HBM)", buf, 8 << 20, 8 << 20, 256);
    return 0;
}
