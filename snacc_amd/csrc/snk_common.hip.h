// snk_common.hip.h -- shared declarations of the gfx950 lz4-frame size kernels.
// (kernels: snk_fast.hip.h, snk_bytes.hip.h, snk_legacy.hip.h, snk_ingest.hip.h; all included by snk_device.hip.h)
//
// Replaces (size only) lz4framed.compress at ref:snacc/pairwise_ncd.py:80 for the
// N + N*N compressions issued by ref:snacc/cli.py:108-129.  Bit-exact against
// liblz4 1.9.3 LZ4F_compressFrame(prefs=NULL); the algorithm statement is in
// SURVEY.md 8(c-spec) and DESIGN.md.
//
// Execution model (CDNA4).  The LZ4 "fast" parse of one stream is a strictly serial chain (each
// probe depends on the previous match length and on every earlier hash-table write), so
// parallelism comes from running MANY independent chains.  One *lane* owns one chain (one ordered
// pair or one single sequence); its hash table lives in LDS, and LDS bytes per chain is what limits
// the chains resident per CU.  Every kernel runs the parse as ONE flat probe loop per lane (search
// probes and post-match probes are the same code; rare events leave through wave-uniform side
// exits) so that lanes in different phases of their parse still execute the same instructions.
//
//   snk_fast.hip.h    2-bit kernel: both sequences pure upper-case ACGT.  Table = 894 collision
//                     classes of liblz4's hash (5-mer -> slot LUT) as 16-bit block offsets + a
//                     "written this block" bitmap: 1904 B per chain, 84 chains per CU.
//   snk_bytes.hip.h   byte kernels for any alphabet: full 4096-slot table (18 chains per CU) or,
//                     when the resident sequences use <= 1024 / 2048 distinct 5-byte hashes, a
//                     renamed compact table (70 / 35 chains per CU); one-shot mode for inputs
//                     <= 64 KiB (13-bit hash of 4 bytes, single block).
//   snk_legacy.hip.h  the first byte kernel (nested loops, liblz4's own u32 / u16 tables), kept as
//                     an independent second implementation for cross-checks.
//   snk_ingest.hip.h  classify / 2-bit pack / hash sets / snapshot conversion.
#pragma once
#ifdef SNK_HOST_EMU                 // g++ build of one lane for the CPU tests (tests/emu/); never shipped
#include "snk_host_emu.h"
#define SNK_AS1
#define SNK_AS3
#define SNK_COOP(stride) 1u         // the one emulated lane walks cooperative loops alone
#else
#include <hip/hip_runtime.h>
#define SNK_AS1 __attribute__((address_space(1)))
#define SNK_AS3 __attribute__((address_space(3)))
#define SNK_COOP(stride) (stride)
#endif
#include <stdint.h>

#define SNK_BLOCK       65536u
#define SNK_MAXDIST     65535u
#define SNK_PAD         64        // zero bytes before and after every sequence buffer
// Zero bytes before the first and after the last sequence of the 2-bit arena: the steady loop of the
// 2-bit kernel forms candidate addresses from table entries before it knows they are in range
// (up to 65 540 bases = 16 385 bytes either side of a sequence), and every such load must stay inside
// the allocation.  Single-sequence jobs use offset SNK_PAD of this zero region as their empty suffix.
#define SNK_ARENA_SLACK 16512

// status bits written by kernels
#define SNK_ST_ITERCAP  1u

struct SnkJob {
    int32_t  xi;        // prefix sequence
    int32_t  yi;        // suffix sequence, -1 = single (stream is x alone)
    uint32_t out_idx;   // where the frame size goes
    int32_t  snap;      // 1 = dump the prefix snapshot of xi when its boundary is reached
};

struct SnkTables {
    // per-sequence, all device pointers
    const uint8_t  *const *bytes;     // ASCII, padded (per-sequence pointers, legacy byte kernel)
    const uint8_t  *bytes_arena;      // the same ASCII data as one allocation < 4 GiB; starts with SNK_PAD zero bytes
    const uint32_t *bytes_off;        // byte offset of each sequence in the ASCII arena
    const uint8_t  *packed_arena;     // 2-bit packed sequences, one allocation < 4 GiB; SNK_ARENA_SLACK zero bytes in front and behind
    const uint8_t  *mask_arena;       // same layout, a 2-bit class per base: 00 the set's letters, 01 the same letters in the other
                                      // case, 11 any other byte (NULL: no resident 2-bit sequence has exceptions)
    const uint32_t *packed_off;       // byte offset of each packed sequence in the arena (0 = not packed)
    const uint32_t *len;
    const uint32_t *snap_pos;         // block-aligned prefix length covered by the snapshot (0 = none)
    uint32_t       *snap_out;         // frame bytes (header included) after snap_pos
    uint32_t       *snap_fast;        // [n][896] slot indexed tables (ACGT sequences), absolute positions
    uint32_t       *snap_gen;         // [n][4096] hash indexed tables
    const uint16_t *lut_slot;         // [1024]  5-mer code -> table slot (0..893); colliding 5-mers share one
    const uint16_t *lut_h2c;          // [4096]  compact byte kernel: hash -> slot 0..CAP-1, 0xFFFF = not in the resident set
    const uint16_t *lut_h2c4;         // [8192]  the same for the one-shot hash of 4 bytes
    const uint16_t *slots;            // slot stream of the byte kernels: [ASCII arena offset + p] = table slot of the 5 bytes at p
                                      // (NULL: every resident pair runs on the 2-bit kernel or in one-shot mode)
    const uint8_t  *zero_pad;         // >= 2*SNK_PAD zero bytes
    // 2-bit kernel on sequences with a few non-ACGT bytes ("exceptions": N runs, IUPAC codes; snk_fast.hip.h)
    const uint32_t *exc_flags;        // 1 bit per 16 bases: an exception lies within bases [16g - 16, 16g + 32)
    const uint32_t *exc_off;          // per sequence: its first word in exc_flags, 0xFFFFFFFF = the sequence has none
    const uint32_t *exc_runs;         // exact runs of non-ACGT bytes, {start, end} pairs sorted by position, each sequence's
                                      // list ends with {0xFFFFFFFF, 0xFFFFFFFF}
    const uint32_t *exc_roff;         // per sequence: index of its first pair in exc_runs (only read when exc_off says it has some)
    const uint16_t *lut_h2s;          // [4096] liblz4 hash of 5 bytes -> slot of the 2-bit table (< 0x8000), or 0x8000 | index of the hash in
                                      //        the chain's overflow table (no 5-mer of the set's case has it)
    const uint16_t *lut_ovi;          // [4096] liblz4 hash -> its index in an overflow table: the hashes of the other case's 5-mers come
                                      //        first, by their compact numbers (lut_oj), then all the others
    const uint16_t *lut_s2h;          // [896]  slot -> hash
    const uint16_t *lut_okey;         // [1024] 5-mer code -> where liblz4 keeps the 5-mer written in the OTHER case: the slot of the
                                      // 2-bit table (< 896) when a 5-mer of the set's case has the same hash, else 0x1000 | index in ovf
    uint32_t       *ovf;              // [resident chains][4096] overflow tables (absolute positions, liblz4's own layout)
    // the other-case mode on a table in LDS (round 4, snk_oth_swap_in / _out): the liblz4 hashes of the 1024 other-case 5-mers
    // numbered 0 .. 894 ("compact other-case slots")
    const uint16_t *lut_oj;           // [1024] 5-mer code -> compact other-case slot (the kernels keep a copy at LDS offset 2048)
    const uint32_t *lut_omap;         // [896]  compact other-case slot j -> liblz4 hash | (slot of the 2-bit table when a 5-mer of the
                                      //        set's case has the same hash, else 0xFFFF) << 16; entry 895 unused.  Entry j of a chain's
                                      //        overflow table belongs to the same hash
    uint32_t       *osave;            // [resident chains][512] where a chain's set-case table waits while its LDS region holds the other case's
    uint32_t        header_bytes;     // 7, or 15 with the content-size field
};

__device__ __forceinline__ uint64_t snk_ld8u(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);        // byte-aligned; gfx950 global/LDS loads allow it
    return v;
}

__device__ __forceinline__ uint32_t snk_lit_ext(uint32_t lit)
{
    return lit >= 15u ? (lit - 15u) / 255u + 1u : 0u;
}

// global-memory (address space 1) pointers keep hipcc on global_load_* instead of flat_load_*
typedef SNK_AS1 const uint8_t snk_g8;
struct __attribute__((packed)) SnkU64 { uint64_t v; };
struct __attribute__((packed)) SnkU32 { uint32_t v; };
__device__ __forceinline__ uint64_t snk_ld8g(snk_g8 *p)
{
    return ((SNK_AS1 const SnkU64 *)p)->v;   // byte-aligned 8-byte load
}
__device__ __forceinline__ uint32_t snk_ld4g(snk_g8 *p)
{
    return ((SNK_AS1 const SnkU32 *)p)->v;   // byte-aligned 4-byte load
}

// ---- byte-accurate view of the concatenation x+y (legacy byte kernel, frame emission, and the 2-bit
// kernel's general path for sequences with exceptions) ----
struct SnkGenSrc {
    const uint8_t *xb, *yb;
    uint32_t lx;
};

__device__ __forceinline__ uint32_t snk_byte_at(const SnkGenSrc &s, uint32_t p)
{
    return p < s.lx ? s.xb[p] : s.yb[p - s.lx];
}

__device__ __forceinline__ uint64_t snk_ld8_straddle(const SnkGenSrc &s, uint32_t p)
{
    uint64_t v = 0;
    for (uint32_t b = 0; b < 8u; ++b) v |= (uint64_t)snk_byte_at(s, p + b) << (8u * b);
    return v;
}

// 8 bytes of the concatenation starting at p (bytes past the end read as padding)
__device__ __forceinline__ uint64_t snk_ld8(const SnkGenSrc &s, uint32_t p)
{
    if (__builtin_expect(p + 8u <= s.lx, 1)) return snk_ld8u(s.xb + p);
    if (p >= s.lx) return snk_ld8u(s.yb + (p - s.lx));
    return snk_ld8_straddle(s, p);
}

__device__ __forceinline__ uint32_t snk_hash5(uint64_t v)
{
    return (uint32_t)(((v << 24) * 889523592379ull) >> 52);
}
__device__ __forceinline__ uint32_t snk_hash4(uint64_t v)
{
    return ((uint32_t)v * 2654435761u) >> 19;
}

// Diagnostic trace (only with -DSNK_TRACE, `make trace`; never shipped): lane 0 of workgroup 0
// appends (tag, a, b, c) records for stream positions >= snk_trace_from.
#ifdef SNK_TRACE
__device__ unsigned int snk_trace_n;
__device__ unsigned int snk_trace_from;
__device__ unsigned int snk_trace_buf[4 * 4096];
__device__ __forceinline__ void snk_trace(unsigned tag, unsigned a, unsigned b, unsigned c, unsigned pos)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && pos >= snk_trace_from) {
        unsigned i = atomicAdd(&snk_trace_n, 1u);
        if (i < 4096u) { snk_trace_buf[4*i] = tag; snk_trace_buf[4*i+1] = a; snk_trace_buf[4*i+2] = b; snk_trace_buf[4*i+3] = c; }
    }
}
#define SNK_TRACE_REC(tag, a, b, c, pos) snk_trace(tag, a, b, c, pos)
#else
#define SNK_TRACE_REC(tag, a, b, c, pos) do { } while (0)
#endif
