// snk_host_emu.h -- TEST INFRASTRUCTURE.  Lets g++ compile the 2-bit kernel sources
// (snacc_amd/csrc/snk_fast.hip.h) for the host and run ONE lane of ONE wave on the CPU, so that the
// kernel's parse logic can be checked against the oracle without a GPU (tests/test_kernel_emu.py).
// A wave of one lane: __any(p) == __all(p) == p, shuffles return their own value, the LDS is a
// static array -- and, inside the two-lane steady loop, a lane PAIR on two host threads in lockstep (round 4).
// Only included when SNK_HOST_EMU is defined; never part of libsnacc_hip.so.
#pragma once
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <thread>

#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))

struct SnkEmuDim { unsigned x, y, z; };
static SnkEmuDim threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {64, 1, 1}, gridDim = {1, 1, 1};

static uint8_t snk_lds8[160 * 1024] __attribute__((aligned(16)));
static unsigned long long snk_emu_oth_trips = 0;      // trips of the steady loop's other-case mode (the tests check that it ran)

// ---- a lane PAIR (snk_fast_steady_spec: two lanes per chain) on two host threads ----------------------------------------
// Outside the two-lane loop the emulation is one lane (role -1).  Inside it the emulated lane is role 0 of its pair and a
// second host thread runs role 1 through the SAME code: the pair's collectives (DPP swap, ballot, __any) and every LDS
// access of the loop are rendezvous points, so that the two threads execute the loop in lockstep, instruction by
// instruction, as the two lanes of a wave do -- the order of the table's reads and writes is the wave's.
static thread_local int snk_emu_role = -1;
struct SnkEmuPairState { std::atomic<unsigned> arrived{0}, sense{0}; uint64_t slot[2] = {0, 0}; };
static SnkEmuPairState snk_emu_pair;
static inline void snk_emu_pair_barrier()
{
    const unsigned s = snk_emu_pair.sense.load(std::memory_order_acquire);
    if (snk_emu_pair.arrived.fetch_add(1, std::memory_order_acq_rel) == 1u) {
        snk_emu_pair.arrived.store(0, std::memory_order_relaxed);
        snk_emu_pair.sense.store(s + 1u, std::memory_order_release);
    } else {
        unsigned spins = 0;
        while (snk_emu_pair.sense.load(std::memory_order_acquire) == s)
            if (++spins > 200u) std::this_thread::yield();
    }
}
static inline uint64_t snk_emu_pair_exchange(uint64_t v)          // the partner's value
{
    snk_emu_pair.slot[snk_emu_role] = v;
    snk_emu_pair_barrier();
    const uint64_t r = snk_emu_pair.slot[snk_emu_role ^ 1];
    snk_emu_pair_barrier();
    return r;
}
template <typename Lane, typename F> static inline void snk_emu_pair_run(F f, Lane &L)
{
    Lane L1;                                 // the partner lane's registers hold nothing of value: it takes the chain's state from role 0
    memset((void *)&L1, 0, sizeof L1);
    std::thread partner([&]() { snk_emu_role = 1; f(true, L1); snk_emu_role = -1; });
    snk_emu_role = 0;
    f(false, L);
    snk_emu_role = -1;
    partner.join();
}
static inline void snk_emu_lds_sync() { if (snk_emu_role >= 0) snk_emu_pair_barrier(); }
// the table of a chain as the two-lane loop sees it: every access a synchronisation point of the pair
template <typename T> struct SnkEmuPtr { T *p; };
template <typename T> struct SnkEmuRef {
    T *p;
    operator T() const { snk_emu_lds_sync(); return __atomic_load_n(p, __ATOMIC_RELAXED); }
    SnkEmuRef &operator=(T v) { snk_emu_lds_sync(); __atomic_store_n(p, v, __ATOMIC_RELAXED); return *this; }
    SnkEmuPtr<T> operator&() const { return SnkEmuPtr<T>{p}; }
};
template <typename T> struct SnkEmuLds {
    T *base;
    explicit SnkEmuLds(T *b) : base(b) {}
    SnkEmuRef<T> operator[](uint32_t i) const { return SnkEmuRef<T>{base + i}; }
};
static inline uint32_t atomicOr(SnkEmuPtr<uint32_t> a, uint32_t v) { snk_emu_lds_sync(); return __atomic_fetch_or(a.p, v, __ATOMIC_RELAXED); }

static inline int __any(int p) { return snk_emu_role < 0 ? p != 0 : ((snk_emu_pair_exchange(p != 0) != 0) | (p != 0)); }
static inline int __all(int p) { return snk_emu_role < 0 ? p != 0 : ((snk_emu_pair_exchange(p != 0) != 0) & (p != 0)); }
static inline unsigned long long snk_emu_ballot(bool p)
{
    if (snk_emu_role < 0) return p ? 1ull : 0ull;
    const unsigned long long o = snk_emu_pair_exchange(p ? 1u : 0u);
    return snk_emu_role == 0 ? ((p ? 1ull : 0ull) | (o << 1)) : (o | ((p ? 1ull : 0ull) << 1));
}
// v_mov_b32_dpp quad_perm:[1,0,3,2] (the only form the emulated code uses): the partner's value inside the pair
static inline int snk_emu_mov_dpp(int v, int ctrl) { return (snk_emu_role < 0 || ctrl != 0xB1) ? v : (int)(uint32_t)snk_emu_pair_exchange((uint32_t)v); }
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) snk_emu_mov_dpp((v), (ctrl))
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
#define __builtin_amdgcn_ballot_w64(p) snk_emu_ballot((p))
