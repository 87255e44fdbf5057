// Dev aid (GPU box): how many bits of an SGPR offset does s_load_dword / s_store_dword take?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/soffset_test tools/micro/soffset_test.hip && /tmp/soffset_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *buf, uint32_t *res)
{
    for (int b = 8; b < 30; ++b) {
        uint32_t off = 1u << b, v = 0x5000u + (uint32_t)b, r;
        asm volatile("s_store_dword %1, %2, %3 glc\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dword %0, %2, %3 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(v), "s"(buf), "s"(off) : "memory");
        if (threadIdx.x == 0) res[b] = r;
    }
}
int main()
{
    const size_t n = (size_t)1 << 30;
    uint32_t *buf, *res;
    if (hipMalloc(&buf, n + 64) != hipSuccess || hipMalloc(&res, 64 * 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, n + 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, res);
    (void)hipDeviceSynchronize();
    static uint32_t h[64]; static uint32_t probe;
    (void)hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 8; b < 30; ++b) {
        (void)hipMemcpy(&probe, (char *)buf + ((size_t)1 << b), 4, hipMemcpyDeviceToHost);
        printf("offset 2^%d: scalar read back 0x%x, memory at that offset 0x%x (expected 0x%x)\n", b, h[b], probe, 0x5000 + b);
    }
    return 0;
}
