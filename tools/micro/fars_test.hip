// Dev aid (GPU box): the memory pattern of a far chain's trip -- get(slot s1), put(slot s2), put(slot s1) on a private
// 896-entry table in global memory -- done with vector loads/stores and with the scalar path (s_load / s_store glc, one lane
// at a time); both must produce the same checksum.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/fars_test tools/micro/fars_test.hip && /tmp/fars_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LANES 16
__global__ void k(uint32_t *tab, unsigned long long *out, int scalar, int iters)
{
    const uint32_t lane = threadIdx.x;
    uint32_t *mine = tab + (size_t)(blockIdx.x * LANES + (lane < LANES ? lane : 0)) * 896;
    const uint32_t gtb = (blockIdx.x * LANES + (lane < LANES ? lane : 0)) * 3584u;
    uint32_t st = 12345u + lane * 977u + blockIdx.x * 131u;
    unsigned long long sum = 0;
    if (lane < LANES) for (int t = 0; t < 896; ++t) mine[t] = 0u;
    __builtin_amdgcn_s_waitcnt(0);
    for (int i = 2; i < iters; ++i) {
        st = st * 1664525u + 1013904223u;
        const uint32_t s1 = (st >> 8) % 894u;
        uint32_t s2 = (st >> 20) % 894u;
        if ((st & 7u) == 0u) s2 = s1;                                  // same slot now and then
        uint32_t e = 0;
        if (!scalar) {
            if (lane < LANES) { e = __hip_atomic_load(mine + s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (s2 != s1) mine[s2] = (uint32_t)i - 2u; mine[s1] = (uint32_t)i; }
        } else {
            const uint32_t a1 = gtb + 4u * s1, a2 = gtb + 4u * (s2 == s1 ? 895u : s2), d2 = (uint32_t)i - 2u, d1 = (uint32_t)i;
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n\t"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_readlane_b32 s[64+\\i], %[a1], \\i\n\t.endr\n\t"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\ts_load_dword s[80+\\i], %[tab], s[64+\\i] glc\n\t.endr\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_writelane_b32 %[e], s[80+\\i], \\i\n\t.endr\n\t"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_readlane_b32 s[80+\\i], %[a2], \\i\n\t.endr\n\t"
                ".irp i,0,1,2,3,4,5,6,7\n\tv_readlane_b32 s[44+(\\i&7)], %[d2], \\i\n\t.endr\n\t"
                ".irp i,0,1,2,3,4,5,6,7\n\ts_store_dword s[44+(\\i&7)], %[tab], s[80+\\i] glc\n\t.endr\n\t"
                ".irp i,8,9,10,11,12,13,14,15\n\tv_readlane_b32 s[44+(\\i&7)], %[d2], \\i\n\t.endr\n\t"
                ".irp i,8,9,10,11,12,13,14,15\n\ts_store_dword s[44+(\\i&7)], %[tab], s[80+\\i] glc\n\t.endr\n\t"
                ".irp i,0,1,2,3,4,5,6,7\n\tv_readlane_b32 s[52+(\\i&7)], %[d1], \\i\n\t.endr\n\t"
                ".irp i,0,1,2,3,4,5,6,7\n\ts_store_dword s[52+(\\i&7)], %[tab], s[64+\\i] glc\n\t.endr\n\t"
                ".irp i,8,9,10,11,12,13,14,15\n\tv_readlane_b32 s[52+(\\i&7)], %[d1], \\i\n\t.endr\n\t"
                ".irp i,8,9,10,11,12,13,14,15\n\ts_store_dword s[52+(\\i&7)], %[tab], s[64+\\i] glc\n\t.endr\n\t"
                : [e] "+&v"(e) : [a1] "v"(a1), [a2] "v"(a2), [d1] "v"(d1), [d2] "v"(d2), [tab] "s"(tab)
                : "memory", "s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59",
                  "s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79",
                  "s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95");
        }
        if (lane < LANES) sum += e * (unsigned long long)(i & 1023);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < LANES) out[blockIdx.x * LANES + lane] = sum;
}
int main()
{
    const int B = 1024, iters = 20000;
    uint32_t *tab; unsigned long long *out;
    if (hipMalloc(&tab, (size_t)B * LANES * 3584 + 4096) != hipSuccess || hipMalloc(&out, B * LANES * 8) != hipSuccess) return 1;
    static unsigned long long h[2][1024 * LANES];
    for (int scalar = 0; scalar < 2; ++scalar) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, tab, out, scalar, iters);
        (void)hipEventRecord(b); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        (void)hipMemcpy(h[scalar], out, sizeof h[0], hipMemcpyDeviceToHost);
        printf("%s path: %.2f ms (%d waves x %d trips)\n", scalar ? "scalar" : "vector", ms, B, iters);
    }
    int bad = 0;
    for (int i = 0; i < B * LANES; ++i) bad += h[0][i] != h[1][i];
    printf("chains whose checksums differ: %d of %d\n", bad, B * LANES);
    return 0;
}
