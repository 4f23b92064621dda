// Dev harness: where do the workgroups of a 512-block, 80-KB-LDS launch land? (XCC, SE, CU, wave slot)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((3 << 11) | 20);    // HW_REG_XCC_ID (id 20), low bits
    }
    // keep the block alive for a while so that all 512 are resident together
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
    if (lds[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int nb = 512;
    unsigned* d; hipMalloc(&d, nb * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 81920, 0, d);
    std::vector<unsigned> h(nb * 2); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    for (int b : {0, 1, 2, 7, 8, 9, 16, 64, 128, 255, 256, 257, 264, 320, 384, 511}) {
        unsigned id = h[b * 2];
        printf("block %3d: xcc %u se %u cu %u simd %u wave_slot %u (hwid %08x)\n", b, h[b * 2 + 1] & 15, (id >> 13) & 7, (id >> 8) & 15, (id >> 4) & 3, id & 15, id);
    }
    // find co-residents of block 0
    for (int b = 1; b < nb; ++b) {
        unsigned a = h[0], c = h[b * 2];
        if (((a >> 8) & 0xff) == ((c >> 8) & 0xff) && (h[1] & 15) == (h[b * 2 + 1] & 15)) printf("co-resident with block 0: block %d (wave slot %u)\n", b, c & 15);
    }
}
