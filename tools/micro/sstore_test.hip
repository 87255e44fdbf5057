// Dev aid (GPU box): do scalar stores / loads with glc see and get seen by vector accesses of the same wave?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/sstore_test tools/micro/sstore_test.hip && /tmp/sstore_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *buf, uint32_t *res, int glc)
{
    const uint32_t lane = threadIdx.x;
    uint32_t *mine = buf + blockIdx.x * 4096;
    // 1) vector store, then scalar load (glc or not) of the same address
    mine[lane] = 1000u + lane;                                   // vector store (write-through to L2)
    __builtin_amdgcn_s_waitcnt(0);
    uint32_t got1 = 0;
    for (int i = 0; i < 64; ++i) {
        uint32_t off = (uint32_t)i * 4u, v;
        if (glc) asm volatile("s_load_dword %0, %1, %2 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(mine), "s"(off) : "memory");
        else     asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(mine), "s"(off) : "memory");
        if (lane == (uint32_t)i) got1 = v;
    }
    // 2) scalar store (glc or not), then vector load (sc1) of the same address
    for (int i = 0; i < 64; ++i) {
        uint32_t off = 1024u + (uint32_t)i * 4u, v = 2000u + (uint32_t)i;
        if (glc) asm volatile("s_store_dword %0, %1, %2 glc" :: "s"(v), "s"(mine), "s"(off) : "memory");
        else     asm volatile("s_store_dword %0, %1, %2" :: "s"(v), "s"(mine), "s"(off) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t got2;
    asm volatile("global_load_dword %0, %1, %2 offset:1024 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got2) : "v"(lane * 4u), "s"(mine) : "memory");
    // 3) scalar store then scalar load (same path)
    uint32_t got3 = 0;
    for (int i = 0; i < 64; ++i) {
        uint32_t off = 2048u + (uint32_t)i * 4u, v = 3000u + (uint32_t)i, r;
        if (glc) asm volatile("s_store_dword %1, %2, %3 glc\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dword %0, %2, %3 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(v), "s"(mine), "s"(off) : "memory");
        else     asm volatile("s_store_dword %1, %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dword %0, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(v), "s"(mine), "s"(off) : "memory");
        if (lane == (uint32_t)i) got3 = r;
    }
    res[(blockIdx.x * 64 + lane) * 3 + 0] = got1;
    res[(blockIdx.x * 64 + lane) * 3 + 1] = got2;
    res[(blockIdx.x * 64 + lane) * 3 + 2] = got3;
}
int main()
{
    const int B = 512;
    uint32_t *buf, *res;
    hipMalloc(&buf, B * 4096 * 4); hipMalloc(&res, B * 64 * 3 * 4);
    for (int glc = 0; glc < 2; ++glc) {
        hipMemset(buf, 0, B * 4096 * 4);
        hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, buf, res, glc);
        hipDeviceSynchronize();
        static uint32_t h[512 * 64 * 3];
        hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost);
        int bad[3] = {0, 0, 0};
        for (int b = 0; b < B; ++b) for (int l = 0; l < 64; ++l) {
            bad[0] += h[(b * 64 + l) * 3 + 0] != 1000u + l;
            bad[1] += h[(b * 64 + l) * 3 + 1] != 2000u + l;
            bad[2] += h[(b * 64 + l) * 3 + 2] != 3000u + l;
        }
        printf("glc=%d: vector store -> scalar load: %d bad; scalar store -> vector sc1 load: %d bad; scalar store -> scalar load: %d bad (of %d)\n",
               glc, bad[0], bad[1], bad[2], B * 64);
    }
    return 0;
}
