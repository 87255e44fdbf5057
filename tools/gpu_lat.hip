// gpu_lat.hip -- dev aid (GPU box): primitive costs in the setting of the 2-bit steady loop
// (4 waves per workgroup, one per SIMD, 21 active lanes, every CU busy):
//   dependent-chain latency of ds_read_u16 / ds_or_rtn_b32 / global_load_dwordx2 (byte-aligned, L1 hit),
//   issue cost of independent and dependent VALU, of s_waitcnt, of SALU.
// Build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/gpu_lat tools/gpu_lat.hip && /tmp/gpu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096

__device__ inline bool i_am_checker(uint32_t tid) { return tid == 0 && blockIdx.x == 0; }

__global__ void k_lat(const uint8_t *arena, uint32_t arena_mask, uint64_t *out, int which, int lanes)
{
    extern __shared__ uint32_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < 40000u; i += blockDim.x) lds[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    if (lane >= (uint32_t)lanes) return;
    uint32_t a = (tid * 1904u + 2048u) & 0x1fffcu, b = tid * 977u, acc = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0);
    if (which == 0) {            // dependent ds_read_u16 chain (address from the previous value)
        for (int i = 0; i < ITERS; ++i) {
            uint32_t v;
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t" : "=v"(v) : "v"(a) : "memory");
            a = ((v * 2u) + tid * 1904u) & 0x1fffeu;
        }
        acc = a;
    } else if (which == 1) {     // dependent ds_or_rtn_b32 chain
        for (int i = 0; i < ITERS; ++i) {
            uint32_t v;
            asm volatile("ds_or_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)\n\t" : "=v"(v) : "v"(a & 0x1fffcu), "v"(1u) : "memory");
            a = ((v * 4u) + tid * 1904u) & 0x1fffcu;
        }
        acc = a;
    } else if (which == 2 || which == 3) {   // dependent global_load_dwordx2 chain, byte-aligned (2) / 8-aligned (3), 16 KiB window per lane set
        uint32_t off = (tid * 37u) & 0x3fffu;
        const uint8_t *base = arena + (size_t)(blockIdx.x & 255u) * 65536u;
        for (int i = 0; i < ITERS; ++i) {
            uint64_t v;
            const uint8_t *p = base + (which == 3 ? (off & ~7u) : off);
            asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)\n\t" : "=v"(v) : "v"(p) : "memory");
            off = ((uint32_t)v + off * 13u + 5u) & 0x3fffu;
        }
        acc = off;
    } else if (which == 4) {     // 64 independent VALU per trip
        for (int i = 0; i < ITERS / 64; ++i)
            asm volatile(".rept 64\n\tv_add_u32_e32 %0, 1, %1\n\t.endr\n\t" : "=v"(acc) : "v"(b));
    } else if (which == 5) {     // 64 dependent VALU per trip
        for (int i = 0; i < ITERS / 64; ++i)
            asm volatile(".rept 64\n\tv_add_u32_e32 %0, 1, %0\n\t.endr\n\t" : "+v"(b));
        acc = b;
    } else if (which == 6) {     // 64 x (VALU + satisfied s_waitcnt)
        for (int i = 0; i < ITERS / 64; ++i)
            asm volatile(".rept 64\n\tv_add_u32_e32 %0, 1, %0\n\ts_waitcnt lgkmcnt(0)\n\t.endr\n\t" : "+v"(b));
        acc = b;
    } else if (which == 7) {     // 64 x (VALU + SALU)
        for (int i = 0; i < ITERS / 64; ++i)
            asm volatile(".rept 64\n\tv_add_u32_e32 %0, 1, %0\n\ts_or_b64 vcc, vcc, vcc\n\t.endr\n\t" : "+v"(b) : : "vcc", "scc");
        acc = b;
    } else if (which == 8) {     // v_cmp -> 2 fillers -> v_cndmask chains (the pattern of the loop)
        for (int i = 0; i < ITERS / 16; ++i)
            asm volatile(".rept 16\n\tv_cmp_lt_u32_e32 vcc, 7, %0\n\tv_add_u32_e32 %1, 1, %1\n\tv_add_u32_e32 %1, 1, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t.endr\n\t"
                         : "+v"(b), "+v"(acc) : : "vcc");
    } else if (which == 9) {     // the loop's LDS group: read_u16 + or_rtn + write + or + write, then wait for the first two
        for (int i = 0; i < ITERS; ++i) {
            uint32_t v, w;
            asm volatile("ds_read_u16 %0, %2\n\tds_or_rtn_b32 %1, %3, %4 offset:1792\n\tds_write_b16 %5, %4\n\tds_or_b32 %3, %4 offset:1800\n\tds_write_b16 %2, %4\n\ts_waitcnt lgkmcnt(3)\n\t"
                         : "=&v"(v), "=&v"(w) : "v"(a & 0x1f7feu), "v"((a & 0x1f000u) + (b & 0x7cu)), "v"(1u), "v"((a + 64u) & 0x1f7feu) : "memory");
            a = ((v ^ w) * 2u + tid * 1904u) & 0x1fffeu; b += 4u;
        }
        acc = a;
    }
    else if (which >= 10 && which <= 13) {
        // the steady loop's load pattern: every lane walks forward ~1.35 B per trip through the SAME 256 KB
        // sequence of its workgroup (lanes within ~2 KB of each other), one candidate window 1..4095 B behind
        // its cursor (dwordx2, byte aligned; 12: dword aligned; 13: distance 1..511 B) and, except in 11, the
        // reservoir refill (dword) just ahead of the cursor; only the candidate window is waited for.
        const uint8_t *base = arena + (size_t)(blockIdx.x & 255u) * 65536u;     // 256 KB apart would exceed the arena: 64 KB stride, 48 KB walk
        uint32_t p = 8192u + (tid * 53u) % 2048u, rnd = tid * 2654435761u + 12345u;
        for (int i = 0; i < ITERS; ++i) {
            rnd = rnd * 1664525u + 1013904223u;
            const uint32_t d = 1u + ((rnd >> 8) & (which == 13 ? 511u : 4095u) & ((rnd >> 24) | 0x1ffu));
            uint32_t off = p - d;
            if (which == 12) off &= ~3u;
            uint64_t v; uint32_t r;
            if (which == 11)
                asm volatile("global_load_dwordx2 %0, %2, %4\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 %1, 0\n\t" : "=&v"(v), "=&v"(r) : "v"(off), "v"(p + 8u), "s"(base) : "memory");
            else
                asm volatile("global_load_dwordx2 %0, %2, %4\n\tglobal_load_dword %1, %3, %4\n\ts_waitcnt vmcnt(1)\n\t" : "=&v"(v), "=&v"(r) : "v"(off), "v"(p + 8u), "s"(base) : "memory");
            p += 1u + (((uint32_t)v ^ rnd) & 1u);                         // data dependent: next addresses need this window
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += r;
        }
        acc += p;
    }
    else if (which == 14 || which == 15) {   // dependent ds_read_b64 chain, byte-aligned (14) / ds_read2_b32 dword-aligned (15): a y window in LDS
        uint32_t off = (tid * 37u) & 0xfffu;
        uint64_t chk = 0;
        for (int i = 0; i < ITERS; ++i) {
            uint64_t v;
            if (which == 14) asm volatile("ds_read_b64 %0, %1 offset:2048\n\ts_waitcnt lgkmcnt(0)\n\t" : "=v"(v) : "v"(off) : "memory");
            else             asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\t" : "=v"(v) : "v"((off & ~3u) + 2048u) : "memory");
            chk ^= v;
            off = ((uint32_t)v + off * 13u + 5u) & 0xfffu;
        }
        acc = off + (uint32_t)chk;
        if (which == 14 && i_am_checker(tid)) {
            // correctness of the unaligned read: compare with byte loads
            uint32_t bad = 0;
            for (uint32_t o = 0; o < 64u; ++o) {
                uint64_t v, w = 0;
                asm volatile("ds_read_b64 %0, %1 offset:2048\n\ts_waitcnt lgkmcnt(0)\n\t" : "=v"(v) : "v"(o) : "memory");
                for (uint32_t b = 0; b < 8u; ++b) w |= (uint64_t)((const uint8_t *)lds)[2048u + o + b] << (8u * b);
                bad += v != w;
            }
            out[2] = bad;
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0 && (tid >> 6) == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}

int main()
{
    uint8_t *arena; uint64_t *out;
    const size_t asz = (size_t)256 * 65536 + 65536;
    hipMalloc((void **)&arena, asz); hipMalloc((void **)&out, 32); hipMemset(out, 0xff, 32);
    std::vector<uint8_t> h(asz);
    for (size_t i = 0; i < asz; ++i) h[i] = (uint8_t)((i * 2654435761ull) >> 13);
    hipMemcpy(arena, h.data(), asz, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_lat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char *name[] = { "ds_read_u16 dependent", "ds_or_rtn_b32 dependent", "global_load_dwordx2 byte-aligned dependent (L1)",
                           "global_load_dwordx2 8-aligned dependent (L1)", "VALU independent", "VALU dependent",
                           "VALU + satisfied s_waitcnt", "VALU + SALU", "v_cmp, 2 VALU, v_cndmask (4 instr)", "loop's LDS group (5 ops, wait for 2)",
                           "loop's loads: window (byte aligned) + refill", "loop's loads: window only", "loop's loads: window dword aligned + refill",
                           "loop's loads: window <= 511 B behind + refill", "ds_read_b64 byte-aligned dependent", "ds_read2_b32 dword-aligned dependent" };
    for (int lanes : { 21 })
        for (int grid : { 1024 })
            for (int w = 0; w < 16; ++w) {
                uint64_t r[3] = { 0, 0, 0 };
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(k_lat, dim3(grid), dim3(256), 160 * 1024, 0, arena, 0u, out, w, lanes);
                    hipDeviceSynchronize();
                }
                hipMemcpy(r, out, 24, hipMemcpyDeviceToHost);
                if (w == 14) printf("   unaligned ds_read_b64 vs byte loads: %llu mismatches of 64\n", (unsigned long long)r[2]);
                const double per = (double)r[0] / ITERS;
                printf("lanes=%2d grid=%4d  %-52s %8.1f ticks/op (x?%s)\n", lanes, grid, name[w], per, "");
            }
    return 0;
}
