// Dev aid (GPU box): where does the dispatcher put the workgroups of a sparse launch?  Every wave records HW_ID / XCC_ID;
// the host counts the distinct compute units used and the most waves any of them got, for several (grid, block, LDS) shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/place_test tools/micro/place_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void spin(unsigned *out, int iters)
{
    extern __shared__ unsigned lds[];
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = x * 3u + 1u;
    }
    if (x == 12345u) lds[threadIdx.x] = x;              // (keeps the LDS allocation alive)
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}
int main()
{
    unsigned *d;
    if (hipMalloc(&d, 8 * 65536) != hipSuccess) return 1;
    const int cfg[][3] = { {256, 256, 0}, {256, 256, 96}, {256, 64, 0}, {1024, 64, 0}, {256, 256, 33}, {256, 256, 63}, {256, 512, 0}, {128, 512, 0}, {256, 256, 160} };
    for (auto &c : cfg) {
        const size_t lds = (size_t)c[2] * 1024;
        if (hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("attr failed\n"); continue; }
        hipLaunchKernelGGL(spin, dim3(c[0]), dim3(c[1]), lds, 0, d, 1 << 16);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); continue; }
        const int waves = c[0] * c[1] / 64;
        std::vector<unsigned> h(2 * waves);
        hipMemcpy(h.data(), d, 8 * waves, hipMemcpyDeviceToHost);
        std::map<unsigned, int> cu, simd;
        for (int w = 0; w < waves; ++w) {
            const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xF;
            const unsigned cuid = (xcc << 16) | (hw & 0xFF00);            // se / sh / cu bits 8..15
            cu[cuid]++; simd[(cuid << 2) | ((hw >> 4) & 3)]++;
        }
        int mx = 0, ms = 0;
        for (auto &kv : cu) mx = kv.second > mx ? kv.second : mx;
        for (auto &kv : simd) ms = kv.second > ms ? kv.second : ms;
        printf("grid %5d x %3d threads, LDS %3d KiB: %4d waves on %3zu CUs (most on one CU: %d), %4zu SIMDs (most on one SIMD: %d); hw_id of wave 0: 0x%08x xcc %u\n",
               c[0], c[1], c[2], waves, cu.size(), mx, simd.size(), ms, h[0], h[1]);
    }
    return 0;
}
