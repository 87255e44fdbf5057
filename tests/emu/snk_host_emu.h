// snk_host_emu.h -- TEST INFRASTRUCTURE.  Lets g++ compile the 2-bit kernel sources
// (snacc_amd/csrc/snk_fast.hip.h) for the host and run ONE lane of ONE wave on the CPU, so that the
// kernel's parse logic can be checked against the oracle without a GPU (tests/test_kernel_emu.py).
// A wave of one lane: __any(p) == __all(p) == p, shuffles return their own value, the LDS is a
// static array.  Only included when SNK_HOST_EMU is defined; never part of libsnacc_hip.so.
#pragma once
#include <stdint.h>
#include <string.h>

#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))

struct SnkEmuDim { unsigned x, y, z; };
static SnkEmuDim threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {64, 1, 1}, gridDim = {1, 1, 1};

static uint8_t snk_lds8[160 * 1024] __attribute__((aligned(16)));
static unsigned long long snk_emu_oth_trips = 0;      // trips of the steady loop's other-case mode (the tests check that it ran)

static inline int __any(int p) { return p != 0; }
static inline int __all(int p) { return p != 0; }
static inline int __shfl(int v, int) { return v; }
static inline void __syncthreads() {}
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
static inline uint32_t atomicAnd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o & v; return o; }

static inline uint32_t snk_emu_alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u));
}
#define __builtin_amdgcn_alignbit(hi, lo, sh) snk_emu_alignbit((hi), (lo), (sh))
#define __builtin_amdgcn_sched_barrier(m) do { } while (0)
#define __builtin_amdgcn_readlane(v, l) (v)
#define __builtin_amdgcn_ballot_w64(p) ((unsigned long long)((p) ? 1ull : 0ull))
