// Dev aid (GPU box): what clock do sparse launches run at?  Every wave runs a fixed chain of dependent VALU instructions and
// reports clock64() (s_memtime) and wall_clock64() (constant 100 MHz) deltas: shader cycles per instruction and the shader
// clock, for launches of `grid` workgroups x `block` threads.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/clk_test tools/micro/clk_test.hip ; run: tools/bin/clk_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long *out, int iters)
{
    unsigned x = threadIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = x * 3u + 1u;       // (v_mad_u32_u24 / v_mul_lo + add: a dependent chain)
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[3 * w] = c1 - c0; out[3 * w + 1] = w1 - w0; out[3 * w + 2] = x;
    }
}
int main()
{
    const int iters = 1 << 20;
    unsigned long long *d;
    hipMalloc(&d, 3 * 8 * 65536);
    const int cfg[][2] = { {256, 64}, {256, 256}, {1024, 64}, {256, 512}, {64, 64}, {1, 64}, {2048, 256}, {256, 64} };
    for (auto &c : cfg) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(spin, dim3(c[0]), dim3(c[1]), 0, 0, d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const int waves = c[0] * c[1] / 64;
        std::vector<unsigned long long> h(3 * waves);
        hipMemcpy(h.data(), d, 3 * 8 * waves, hipMemcpyDeviceToHost);
        double sc = 0, sw = 0;
        for (int w = 0; w < waves; ++w) { sc += h[3 * w]; sw += h[3 * w + 1]; }
        sc /= waves; sw /= waves;
        printf("grid %5d x %3d threads: kernel %.2f ms; per wave: clock64 %.0f, wall_clock64 %.0f (%.2f ms at 100 MHz) -> %.3f GHz; clock64 ticks per instr (32 per iter) %.2f\n",
               c[0], c[1], ms, sc, sw, sw / 1e5, sc / (sw * 10.0), sc / ((double)iters * 32));
    }
    return 0;
}
