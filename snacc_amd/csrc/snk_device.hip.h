// snk_device.hip.h -- gfx950 device code for the lz4-frame size kernels.
//
// Replaces (size only) lz4framed.compress at ref:snacc/pairwise_ncd.py:80 for the
// N + N*N compressions issued by ref:snacc/cli.py:108-129.  Bit-exact against
// liblz4 1.9.3 LZ4F_compressFrame(prefs=NULL); the algorithm statement is in
// SURVEY.md 8(c-spec) and DESIGN.md.
//
// Execution model (CDNA4): the LZ4 "fast" parse of one stream is a strictly
// serial chain (each probe depends on the previous match length and on every
// earlier hash-table write), so parallelism comes from running MANY independent
// chains.  One *lane* owns one chain (one ordered pair); its hash table lives in
// LDS.  LDS bytes per chain is the occupancy limiter, hence two kernels:
//
//   snk_fast_kernel : both sequences are pure upper-case ACGT.  Sequences are
//       2-bit packed; the 12-bit hash of 5 bytes only ever sees 1024 5-mers, so
//       the table is indexed by the 10-bit 5-mer code (4 KiB per chain instead
//       of 16 KiB).  5-mers that collide in liblz4's hash share a slot there;
//       here a put() writes the colliding partners too (LUT, <=3 partners).
//   snk_generic_kernel : any bytes.  4096 x u32 table (linked mode) or
//       8192 x u16 (one-shot mode for inputs <= 64 KiB), 16 KiB per chain.
//
// Both run the parse as ONE flat probe loop per lane (search probes and
// post-match probes are the same code) so that lanes of a wave that are in
// different phases of their parse still execute the same instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SNK_BLOCK       65536u
#define SNK_MAXDIST     65535u
#define SNK_PAD         64        // zero bytes before and after every sequence buffer

// status bits written by kernels
#define SNK_ST_ITERCAP  1u
#define SNK_ST_BADJOB   2u

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
    const uint8_t  *packed_arena;     // 2-bit packed sequences, one allocation < 4 GiB; starts with 4*SNK_PAD zero bytes
    const uint32_t *packed_off;       // byte offset of each packed sequence in the arena (0 = not packed)
    const uint32_t *len;
    const uint32_t *snap_pos;         // block-aligned prefix length covered by the snapshot (0 = none)
    uint32_t       *snap_out;         // frame bytes (header included) after snap_pos
    uint32_t       *snap_fast;        // [n][896] slot indexed tables (ACGT sequences), absolute positions
    uint32_t       *snap_gen;         // [n][4096] hash indexed tables
    const uint16_t *lut_slot;         // [1024]  5-mer code -> table slot (0..893); colliding 5-mers share one
    const uint16_t *lut_h2c;          // [4096]  compact byte kernel: hash -> slot 0..CAP-1, 0xFFFF = not in the resident set
    const uint16_t *lut_h2c4;         // [8192]  the same for the one-shot hash of 4 bytes
    const uint8_t  *zero_pad;         // >= 2*SNK_PAD zero bytes
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

// =========================================================================
//  2-bit ACGT kernel
// =========================================================================
//
// LDS per chain (1904 B): tbl[896] u16 + bm[28] u32.
//   liblz4's table maps a 12-bit hash slot to the last inserted absolute position
//   and rejects candidates further than 65535 back.  Pure-ACGT input reaches only
//   894 slots (1024 5-mers, colliding ones share a slot: LUT `slot[5-mer]`).
//   Positions are kept as 16-bit offsets inside their 64 KiB frame block, with one
//   bit per slot saying "written during the current block":
//     bit set            -> candidate = block_base + off            (distance < 64 Ki)
//     bit clear, off > c -> candidate = block_base - 65536 + off    (previous block,
//                           distance = 65536 + c - off <= 65535 exactly when off > c)
//     otherwise          -> too far / never written (off 0 is never > c)
//   At every block transition entries whose bit is clear (older than one block)
//   are zeroed and the bitmap is cleared: exactly liblz4's "too far" rule.
#define SNK_FSLOTS      896u                    // 894 used, padded to a multiple of 32
#define SNK_FBMWORDS    28u
#define SNK_FCHAIN_B    (SNK_FSLOTS * 2u + SNK_FBMWORDS * 4u)      // 1904 bytes
#define SNK_FLUT_B      2048u                   // slot LUT: 1024 x u16

// global-memory (address space 1) pointers keep hipcc on global_load_* instead of flat_load_*
typedef __attribute__((address_space(1))) const uint8_t snk_g8;
struct __attribute__((packed)) SnkU64 { uint64_t v; };
struct __attribute__((packed)) SnkU32 { uint32_t v; };
__device__ __forceinline__ uint64_t snk_ld8g(snk_g8 *p)
{
    return ((__attribute__((address_space(1))) const SnkU64 *)p)->v;   // byte-aligned 8-byte load
}
__device__ __forceinline__ uint32_t snk_ld4g(snk_g8 *p)
{
    return ((__attribute__((address_space(1))) const SnkU32 *)p)->v;   // byte-aligned 4-byte load
}

struct SnkFastSrc {
    snk_g8 *arena;            // wave-uniform base of the packed arena (SGPR base + 32-bit lane offsets)
    uint32_t xoff, yoff;      // byte offsets of the two packed sequences inside the arena
    uint32_t lx;
};

// 32-bit window: the 16 bases [q, q+16) of the sequence at arena offset `off`, base q at bits 0..1.
__device__ __forceinline__ uint32_t snk_w32_at(snk_g8 *arena, uint32_t off, int32_t q)
{
    const uint64_t v = snk_ld8g(arena + (size_t)(uint32_t)((int32_t)off + (q >> 2)));
    return __builtin_amdgcn_alignbit((uint32_t)(v >> 32), (uint32_t)v, (uint32_t)(q & 3) * 2u);
}

// Window over the virtual concatenation x+y at stream position p: bases [p-4, p+12).
// 5-mer at p = bits 8..17, 5-mer at p-2 = bits 4..13.
__device__ __forceinline__ uint32_t snk_fetch32(const SnkFastSrc &s, uint32_t p)
{
    const int32_t q0 = (int32_t)p - 4;
    const bool inx = (p + 12u <= s.lx);
    const bool iny = (q0 >= (int32_t)s.lx);
    if (__builtin_expect(inx | iny, 1))
        return snk_w32_at(s.arena, iny ? s.yoff : s.xoff, iny ? q0 - (int32_t)s.lx : q0);
    // seam: q0 < lx < q0 + 16.  x is zero padded beyond lx.
    const uint32_t xv = snk_w32_at(s.arena, s.xoff, q0);
    const uint32_t sh = 2u * (uint32_t)((int32_t)s.lx - q0);        // 2..30
    return xv | ((uint32_t)snk_ld8g(s.arena + (size_t)s.yoff) << sh);
}

__device__ __forceinline__ uint32_t snk_base_at(const SnkFastSrc &s, uint32_t p)
{
    return snk_fetch32(s, p + 4u) & 3u;
}

// Cursor-side reservoir: 32 packed bases [rb, rb+32) of ONE source sequence in registers
// (r0, r1) plus the next 16 (nx) already in flight, so the window at the probe position costs
// no memory latency.  (rb - org) % 4 == 0.  lim = 0 marks "no usable window".
struct SnkWin {
    uint32_t soff;     // arena offset of the source sequence
    uint32_t org;      // stream position of base 0 of the source (0 for x, lx for y)
    uint32_t rb;       // stream position of bit 0 of r0
    uint32_t lim;      // largest probe position this source can serve
    uint32_t r0, r1, nx;
};

__device__ __forceinline__ void snk_win_init(SnkWin &w, snk_g8 *arena, uint32_t soff, uint32_t org,
                                             uint32_t lim, uint32_t cur)
{
    w.soff = soff; w.org = org; w.lim = lim;
    w.rb = org + ((cur - 4u - org) & ~3u);
    snk_g8 *p = arena + (size_t)(soff + ((w.rb - org) >> 2));
    w.r0 = snk_ld4g(p); w.r1 = snk_ld4g(p + 4); w.nx = snk_ld4g(p + 8);
}

// Everything one lane (= one chain = one ordered pair) carries through the flat parse loop.
struct SnkFastLane {
    SnkFastSrc s;
    uint32_t n, spos;
    int32_t xi, snap;
    uint32_t out_idx;
    // progress over the frame
    uint32_t pos, total, iend, blen, blocks_left;
    bool first, in_block;
    // parse state inside the current block
    uint32_t cur, step, nb, anchor, op;
    uint32_t mfl1, mlimit, olimit, base;   // base = stream position of the block start
    uint32_t endcode;                      // 0 running, 1 ends with last-literals, 2 liblz4 gave up (raw)
    bool pending;                          // put(cur-2) owed before the next probe
    bool yflag;                            // whole block lies > 64 KiB + 8 past the seam, window on y
    SnkWin w;
};

// Rare path (once per 64 KiB): close the finished block, age the table, open the next block.
// Returns true when the frame is complete (size written).
__device__ __forceinline__ bool snk_fast_block_step(SnkFastLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                                 const uint16_t *slot, uint32_t *out, uint32_t *status)
{
    if (L.in_block) {
        uint32_t payload = L.blen;
        if (L.endcode != 2u) {
            const uint32_t run = L.iend - L.anchor;
            if (L.op + run + 1u + (run + 240u) / 255u <= L.olimit)
                payload = L.op + 1u + snk_lit_ext(run) + run;
        }
        L.total += 4u + payload;
        L.pos = L.iend;
        L.in_block = false;
        L.endcode = 0u;
    }
    for (;;) {
        if (L.snap != 0 && L.pos == L.spos && L.spos != 0u) {
            // prefix snapshot: absolute positions; entries older than one block -> 0 (too far for good)
            uint32_t *dst = T.snap_fast + (size_t)L.xi * SNK_FSLOTS;
            for (uint32_t t = 0; t < SNK_FSLOTS; ++t)
                dst[t] = ((bm[t >> 5] >> (t & 31u)) & 1u) ? (L.pos - 65536u + tbl[t]) : 0u;
            T.snap_out[L.xi] = L.total;
        }
        if (L.pos >= L.n) { out[L.out_idx] = L.total + 4u; return true; }      // + end mark
        if (L.blocks_left-- == 0u) { atomicOr(status, SNK_ST_ITERCAP); return true; }
        L.blen = L.n - L.pos < SNK_BLOCK ? L.n - L.pos : SNK_BLOCK;
        L.iend = L.pos + L.blen;
        if (L.blen < 13u) {                         // always stored raw; table untouched
            L.total += 4u + L.blen;
            L.pos = L.iend;
            continue;
        }
        if (!L.first) {
            // age the table: entries not written during the block just finished are dead
            for (uint32_t wi = 0; wi < SNK_FBMWORDS; ++wi) {
                uint32_t z = ~bm[wi];
                while (z) {
                    const uint32_t b = (uint32_t)__builtin_ctz(z);
                    tbl[wi * 32u + b] = 0;
                    z &= z - 1u;
                }
                bm[wi] = 0u;
            }
        }
        L.first = false;
        L.base = L.pos;
        L.mfl1 = L.iend - 11u; L.mlimit = L.iend - 5u; L.olimit = L.blen - 1u;
        {
            const uint32_t w0 = snk_fetch32(L.s, L.pos);
            const uint32_t s0 = slot[(w0 >> 8) & 1023u];
            tbl[s0] = 0;                                              // offset 0 of this block
            atomicOr(&bm[s0 >> 5], 1u << (s0 & 31u));
        }
        L.cur = L.pos + 1u; L.step = 1u; L.nb = 64u; L.anchor = L.pos; L.op = 0u;
        L.pending = false; L.in_block = true;
        if (L.cur >= L.s.lx + 4u) snk_win_init(L.w, L.s.arena, L.s.yoff, L.s.lx, 0xFFFFFFFFu, L.cur);
        L.yflag = L.pos >= L.s.lx + SNK_BLOCK + 8u;
        return false;
    }
}

// Rare path of a match: long back-extension, long match, length-extension bytes, output
// budget, end of block -- liblz4's exact accounting.
__device__ __forceinline__ void snk_fast_match_slow(SnkFastLane &L, uint32_t cur, uint32_t cand, uint32_t f,
                                                    uint32_t anchor0, uint32_t op0)
{
    const SnkFastSrc &s = L.s;
    uint32_t ip = cur, lit = cur - anchor0;
    while (ip > anchor0 && cand > 0u && snk_base_at(s, ip - 1u) == snk_base_at(s, cand - 1u)) { ip--; cand--; lit--; }
    uint32_t e2 = cur + f;
    if (f == 12u) {                                          // keep counting, 16 bases at a time
        uint32_t bpos = cand + (cur - ip) + 12u;
        while (e2 < L.mlimit) {
            const uint32_t d = snk_fetch32(s, e2 + 4u) ^ snk_fetch32(s, bpos + 4u);
            const uint32_t cnt = d ? ((uint32_t)__builtin_ctz(d) >> 1) : 16u;
            e2 += cnt; bpos += cnt;
            if (cnt < 16u) break;
        }
    }
    if (e2 > L.mlimit) e2 = L.mlimit;
    const uint32_t mc = e2 - ip - 4u;
    uint32_t op = op0 + 1u;
    bool bail = op + lit + 8u + lit / 255u > L.olimit;
    if (!bail) {
        op += lit + snk_lit_ext(lit) + 2u;
        bail = op + 6u + (mc + 240u) / 255u > L.olimit;
        if (mc >= 15u) op += (mc - 15u) / 255u + 1u;
    }
    if (bail) { L.endcode = 2u; L.mfl1 = 0u; L.step = 1u; L.cur = cur; L.anchor = anchor0; L.op = op0; return; }
    L.op = op;
    L.anchor = e2;
    L.cur = e2; L.step = 1u; L.nb = 63u; L.pending = true;
    if (e2 >= L.mfl1) { L.endcode = 1u; L.mfl1 = 0u; }
}

// One iteration of the flat parse loop = one probe (search probes and post-match probes are the
// same code).  Returns true when the lane's frame is complete.
// YONLY: wave-uniform promise that every lane is in a block lying > 64 KiB + 8 past its seam with
// its reservoir on y, so cursor and candidate windows both come from y.
// The common path is branch-free (selects); everything rare funnels into two branches.
template <bool YONLY>
__device__ __forceinline__ bool snk_fast_iter(SnkFastLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                              const uint16_t *slot, uint32_t *out, uint32_t *status)
{
    const uint32_t cur = L.cur;
    const uint32_t next = cur + L.step;
    SnkWin &w = L.w;

    // ---- cursor reservoir: slide by 16 bases when needed, refill always in flight ----
    uint32_t o = cur - 4u - w.rb;                            // need 0 <= o <= 15
    {
        const bool sl = (o - 16u) < 16u;
        w.r0 = sl ? w.r1 : w.r0;
        w.r1 = sl ? w.nx : w.r1;
        w.rb += sl ? 16u : 0u;
        o -= sl ? 16u : 0u;
    }
    bool wslow = false;
    if (__builtin_expect((next > L.mfl1) | (o > 15u) | (!YONLY && cur > w.lim), 0)) {
        if (next > L.mfl1)                                   // block end, bail-out, or not started yet
            return snk_fast_block_step(L, T, tbl, bm, slot, out, status);
        // long jump / source change / seam: re-seat the reservoir
        if (cur >= L.s.lx + 4u)                          snk_win_init(w, L.s.arena, L.s.yoff, L.s.lx, 0xFFFFFFFFu, cur);
        else if (cur >= 4u && cur + 12u <= L.s.lx)       snk_win_init(w, L.s.arena, L.s.xoff, 0u, L.s.lx - 12u, cur);
        else                                             { w.lim = 0u; wslow = true; }
        o = cur - 4u - w.rb;
    }
    const uint32_t wc = (!YONLY && wslow) ? snk_fetch32(L.s, cur) : __builtin_amdgcn_alignbit(w.r1, w.r0, 2u * o);

    // ---- table probe: two LDS round trips (slot LUT, then table + bitmap) ----
    const uint32_t s1 = slot[(wc >> 8) & 1023u];             // slot of the 5-mer at cur
    uint32_t s2 = slot[(wc >> 4) & 1023u];                   // slot of the 5-mer at cur-2
    s2 = L.pending ? s2 : (SNK_FSLOTS - 1u);                 // nothing owed: aim the put at the unused slot
    const uint32_t e = tbl[s1];
    const uint32_t bw = bm[s1 >> 5];
    const uint32_t c = cur - L.base;                         // offset in the block, 1..65535
    const uint32_t bit1 = 1u << (s1 & 31u);
    const bool iscur = (bw & bit1) != 0u;
    uint32_t cand = L.base + e - (iscur ? 0u : 65536u);
    bool valid = iscur | (e > c);
    // liblz4 puts cur-2 BEFORE it reads the slot of cur: same slot => the candidate is cur-2
    const bool same = (s2 == s1);
    cand = same ? cur - 2u : cand;
    valid |= same;
    tbl[s2] = (uint16_t)(c - 2u);
    atomicOr(&bm[s2 >> 5], 1u << (s2 & 31u));
    tbl[s1] = (uint16_t)c;
    atomicOr(&bm[s1 >> 5], bit1);
    const uint32_t s3 = L.nb >> 6;
    const uint32_t nstep = s3 ? s3 : 1u;
    cand = valid ? cand : cur;                               // keep the fetch in bounds

    // ---- candidate window: the one global-memory round trip of the probe ----
    // The reservoir refill is issued right next to it (same address again when nothing slid), so
    // both loads are in flight together and neither is waited for alone.
    __builtin_amdgcn_sched_barrier(0);
    snk_g8 *nxp = L.s.arena + (size_t)(w.soff + ((w.rb + 32u - w.org) >> 2));
    const uint32_t wd = YONLY ? snk_w32_at(L.s.arena, L.s.yoff, (int32_t)(cand - 4u - L.s.lx))
                              : snk_fetch32(L.s, cand);
    w.nx = snk_ld4g(nxp);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t x = wc ^ wd;
    const uint32_t f = (uint32_t)__builtin_ctz((x >> 8) | (1u << 24)) >> 1;      // equal bases from cur, 0..12
    const bool m = valid & (f >= 4u);

    // ---- match bookkeeping, computed for every lane and committed by select ----
    uint32_t lit = cur - L.anchor;
    const uint32_t eq = (uint32_t)__builtin_clz(((x & 0xFFu) << 24) | 0x00800000u) >> 1;   // equal bases before cur, 0..4
    uint32_t b = eq < lit ? eq : lit;
    b = b < cand ? b : cand;
    lit -= b;
    uint32_t e2 = cur + f;
    e2 = e2 < L.mlimit ? e2 : L.mlimit;
    const uint32_t mc = e2 - (cur - b) - 4u;
    const uint32_t opn = L.op + lit + 3u;                    // token + literals + offset when no extension bytes
    const uint32_t big = lit > mc ? lit : mc;
    // both limitedOutput checks of liblz4 reduce to op + lit + 9 > olimit when lit, mc < 15
    const bool rare = m & ((b == 4u) | (f == 12u) | (big >= 15u) | (opn + 6u > L.olimit) | (e2 >= L.mfl1));
    if (__builtin_expect(rare, 0)) {
        snk_fast_match_slow(L, cur, cand, f, L.anchor, L.op);
        return false;
    }
    L.op = m ? opn : L.op;
    L.anchor = m ? e2 : L.anchor;
    L.cur = m ? e2 : next;
    L.step = m ? 1u : nstep;
    L.nb = m ? 63u : L.nb + 1u;
    L.pending = m;
    return false;
}

// Candidate window for the seam-aware tight loop, branch-free: one window from x and one from y
// are always in flight together and combined by selects (a straddling window is x's zero-padded
// tail OR-ed with y's head shifted into place).
__device__ __forceinline__ uint32_t snk_fetch32_nobranch(const SnkFastSrc &s, uint32_t p)
{
    const int32_t q0 = (int32_t)p - 4;
    const bool inx = (p + 12u <= s.lx);
    const bool iny = (q0 >= (int32_t)s.lx);
    const uint32_t xv = snk_w32_at(s.arena, s.xoff, iny ? 0 : q0);
    const uint32_t yv = snk_w32_at(s.arena, s.yoff, iny ? q0 - (int32_t)s.lx : 0);
    const uint32_t sh = 2u * (uint32_t)((int32_t)s.lx - q0);        // 2..30 when straddling
    const uint32_t mix = xv | (yv << (sh & 31u));
    return iny ? yv : (inx ? xv : mix);
}

#ifdef SNK_STAMP
__device__ unsigned long long snk_stamp_buf[8];     // diagnostic build only; read by snk_debug_read_stamps
#define SNK_STAMP_T(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SNK_STAMP_T(v) do { } while (0)
#endif
#define SNK_LOOP_DONE   0
#define SNK_LOOP_SWITCH 1

// The parse as a tight loop.  The body has no divergent branch: rare events are detected per lane
// and the wave takes a UNIFORM side exit (__any) to serve them.  The loop is rotated: the slot-LUT
// reads of the NEXT probe are issued as soon as the match length is known, and this probe's
// bookkeeping runs in their shadow.
//   YONLY = true : every active lane's block lies > 64 KiB + 8 past its seam and its reservoir is
//                  on y (98 % of the probes of a 1 Mbp pair); lanes leave only by finishing.
//   YONLY = false: seam-aware candidate fetch; lanes whose reservoir cannot serve the cursor (seam,
//                  stream start) are stepped by the general one-probe routine in the side exit.
//                  Returns SNK_LOOP_SWITCH (wave-uniform) once every active lane has yflag.
// Invariant at the head: w.nx holds the bases [rb+32, rb+48).
template <bool YONLY>
__device__ __forceinline__ int snk_fast_loop(SnkFastLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                             const uint16_t *slot, uint32_t *out, uint32_t *status)
{
    SnkWin &w = L.w;
    snk_g8 *const arena = L.s.arena;
    const uint32_t ybias = L.s.lx + 4u;          // candidate window of stream position p starts at y base p - ybias
    // The slot LUT sits at LDS address 0 (the kernel has no static LDS; the host checks it):
    // indexing it from a constant base saves the per-read base addition.
    (void)slot;
#ifdef SNK_STAMP
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    unsigned long long acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0, iters = 0;
#endif
    const __attribute__((address_space(3))) uint16_t *const lut0 = (const __attribute__((address_space(3))) uint16_t *)0;

    for (;;) {
        // ======== head: serve rare pre-conditions, then start the LUT reads ========
        uint32_t cur, next, o;
        for (;;) {
            cur = L.cur;
            next = cur + L.step;
            o = cur - 4u - w.rb;
            const bool pre = (next > L.mfl1) | (o > 15u) | (!YONLY && cur > w.lim);
            if (__builtin_expect(!__any(pre), 1)) break;
            if (pre) {
                if (YONLY) {
                    if (next > L.mfl1) {                 // block end / bail-out
                        if (snk_fast_block_step(L, T, tbl, bm, slot, out, status)) return SNK_LOOP_DONE;
                    } else {                             // long jump: re-seat the reservoir on y
                        snk_win_init(w, arena, L.s.yoff, L.s.lx, 0xFFFFFFFFu, cur);
                    }
                } else {
                    // one fully general probe: opens/closes blocks, re-seats the reservoir, walks the seam
                    if (snk_fast_iter<false>(L, T, tbl, bm, slot, out, status)) return SNK_LOOP_DONE;
                }
            }
            if (!YONLY && __all(L.yflag)) return SNK_LOOP_SWITCH;
        }
        uint32_t wc = __builtin_amdgcn_alignbit(w.r1, w.r0, 2u * o);
        uint32_t s1 = lut0[(wc >> 8) & 1023u];
        uint32_t s2e = L.pending ? (uint32_t)lut0[(wc >> 4) & 1023u] : (SNK_FSLOTS - 1u);   // nothing owed: unused slot
        uint32_t nxoff = w.soff + ((w.rb + 32u - w.org) >> 2);      // arena offset of the bases [rb+32, rb+48)
        const uint32_t olim6 = L.olimit - 6u;                       // olimit >= 12 inside an open block

        // ======== steady state: one probe per trip, LUT reads for the next one already in flight ========
        for (;;) {
            SNK_STAMP_T(t0);
            const uint32_t c = cur - L.base;
            const uint32_t bit1 = 1u << (s1 & 31u);
            // liblz4's order: put(cur-2), then read the slot of cur, then put(cur).  The LDS executes a
            // wave's operations in issue order, so a put to the same slot is seen by the read.
            // Only the cheap 16-bit write of the owed put goes in front of the read; its bitmap bit
            // follows the read and is patched in by one compare.
            tbl[s2e] = (uint16_t)(c - 2u);
            const uint32_t e = tbl[s1];
            const uint32_t bw = bm[s1 >> 5];
            atomicOr(&bm[s2e >> 5], 1u << (s2e & 31u));
            tbl[s1] = (uint16_t)c;
            atomicOr(&bm[s1 >> 5], bit1);
            const bool hit = (bw & bit1) != 0u;
            const bool same = (s2e == s1);
            const bool iscur = hit || same;
            const bool valid = iscur || (e > c);
            uint32_t cand = (iscur ? L.base : L.base - 65536u) + e;
            cand = valid ? cand : cur;
#ifdef SNK_STAMP
            asm volatile("" :: "v"(cand));
            SNK_STAMP_T(t1);                                  // table data arrived, candidate known
#endif
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t wd = YONLY ? snk_w32_at(arena, L.s.yoff, (int32_t)(cand - ybias))
                                      : snk_fetch32_nobranch(L.s, cand);
            w.nx = snk_ld4g(arena + (size_t)nxoff);
            __builtin_amdgcn_sched_barrier(0);

            const uint32_t x = wc ^ wd;
#ifdef SNK_STAMP
            asm volatile("" :: "v"(x));
            __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0)
            SNK_STAMP_T(t2);                                  // candidate window arrived
#endif
            const uint32_t f = (uint32_t)__builtin_ctz((x >> 8) | (1u << 24)) >> 1;
            const bool m = valid & (f >= 4u);
            uint32_t e2 = cur + f;
            e2 = e2 < L.mlimit ? e2 : L.mlimit;
            const uint32_t s3 = L.nb >> 6;
            const uint32_t nstep = m ? 1u : (s3 ? s3 : 1u);
            const uint32_t ncur = m ? e2 : next;

            // ---- next probe: reservoir + LUT reads (issued before this probe's bookkeeping) ----
            const uint32_t nnext = ncur + nstep;
            uint32_t no = ncur - 4u - w.rb;
            const bool sl = (no - 16u) < 16u;
            const uint32_t sl16 = sl ? 16u : 0u;
            const uint32_t r0n = sl ? w.r1 : w.r0;       // w.nx (just refilled, waited for with wd)
            const uint32_t r1n = sl ? w.nx : w.r1;
            no -= sl16;
            const uint32_t nwc = __builtin_amdgcn_alignbit(r1n, r0n, 2u * (no & 15u));
            const uint32_t ns1 = lut0[(nwc >> 8) & 1023u];
            const uint32_t ns2 = lut0[(nwc >> 4) & 1023u];
            __builtin_amdgcn_sched_barrier(0);           // keep the LUT reads in front of the bookkeeping
#ifdef SNK_STAMP
            SNK_STAMP_T(t3);                                  // next LUT reads issued
#endif

            // ---- bookkeeping of this probe, in the shadow of the LUT reads ----
            const uint32_t anchor0 = L.anchor, op0 = L.op;
            uint32_t lit = cur - anchor0;
            const uint32_t eq = (uint32_t)__builtin_clz(((x & 0xFFu) << 24) | 0x00800000u) >> 1;
            uint32_t b = eq < lit ? eq : lit;
            b = b < cand ? b : cand;
            lit -= b;
            const uint32_t mc = e2 - (cur - b) - 4u;
            const uint32_t opn = op0 + lit + 3u;
            // rare: back-extension reaches 4 (b+11 >= 15), match reaches 12 (f+3 >= 15), a length needs
            // extension bytes (>= 15), or the output budget is at risk.  (A match that ends the block
            // needs no special case: the head closes the block from the committed op/anchor.)
            uint32_t mx = lit > mc ? lit : mc;
            { const uint32_t t1 = b + 11u, t2 = f + 3u; const uint32_t t3 = t1 > t2 ? t1 : t2; mx = mx > t3 ? mx : t3; }
            const bool rare = m & ((mx >= 15u) | (opn > olim6));
            const bool pre = (nnext > L.mfl1) | (no > 15u) | (!YONLY && ncur > w.lim);
            L.op = m ? opn : op0;
            L.anchor = m ? e2 : anchor0;
            L.step = nstep;
            L.nb = m ? 63u : L.nb + 1u;
            w.r0 = r0n; w.r1 = r1n; w.rb += sl16; nxoff += sl16 >> 2;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare | pre) != 0ull, 0)) {
                L.cur = ncur; L.pending = m;
                if (rare) snk_fast_match_slow(L, cur, cand, f, anchor0, op0);
                // restore the head invariant: the reservoir may just have slid
                w.nx = snk_ld4g(arena + (size_t)nxoff);
                break;                                   // the head re-derives everything from L
            }
            cur = ncur; next = nnext; wc = nwc; s1 = ns1; s2e = m ? ns2 : (SNK_FSLOTS - 1u);
#ifdef SNK_STAMP
            asm volatile("" :: "v"(s1), "v"(s2e));            // forces the LUT data to have arrived
            SNK_STAMP_T(t4);
            acc1 += t1 - t0; acc2 += t2 - t1; acc3 += t3 - t2; acc4 += t4 - t3; iters++;
            if (YONLY && blockIdx.x == 0 && threadIdx.x == 0 && (iters & 1023) == 0) {
                snk_stamp_buf[0] = acc1; snk_stamp_buf[1] = acc2; snk_stamp_buf[2] = acc3; snk_stamp_buf[3] = acc4; snk_stamp_buf[4] = iters;
            }
#endif
        }
    }
}

// One chain of the 2-bit kernel.  `lds` = this chain's 1904 bytes, `slot` = the
// workgroup's 5-mer -> slot LUT.
__device__ __forceinline__ void snk_fast_chain(const SnkTables &T, const SnkJob job,
                                               uint8_t *lds, const uint16_t *slot,
                                               uint32_t *out, uint32_t *status)
{
    uint16_t *tbl = (uint16_t *)lds;
    uint32_t *bm = (uint32_t *)(lds + SNK_FSLOTS * 2u);
    SnkFastLane L;
    const uint32_t lx = T.len[job.xi];
    const uint32_t ly = job.yi >= 0 ? T.len[job.yi] : 0u;
    L.s.arena = (snk_g8 *)T.packed_arena;
    L.s.xoff = T.packed_off[job.xi];
    L.s.yoff = job.yi >= 0 ? T.packed_off[job.yi] : SNK_PAD;         // zero region at the arena start
    L.s.lx = lx;
    L.n = lx + ly;
    L.spos = T.snap_pos[job.xi];
    L.xi = job.xi; L.snap = job.snap; L.out_idx = job.out_idx;
    if (job.snap == 0 && L.spos != 0u) { L.pos = L.spos; L.total = T.snap_out[job.xi]; }
    else                               { L.pos = 0u;     L.total = T.header_bytes; }
    L.blocks_left = (L.n >> 16) + 4u;
    L.iend = 0; L.blen = 0; L.first = true; L.in_block = false;
    L.cur = 0; L.step = 1; L.nb = 64; L.anchor = 0; L.op = 0;
    L.mfl1 = 0; L.mlimit = 0; L.olimit = 0; L.base = L.pos; L.endcode = 0;     // mfl1 = 0: first iteration opens a block
    L.pending = false; L.yflag = false;
    L.w.soff = L.s.xoff; L.w.org = 0u; L.w.rb = 0u; L.w.lim = 0u; L.w.r0 = L.w.r1 = L.w.nx = 0u;

    for (;;) {
        int r;
        if (__all(L.yflag)) r = snk_fast_loop<true>(L, T, tbl, bm, slot, out, status);    // deep inside y, to the end
        else                r = snk_fast_loop<false>(L, T, tbl, bm, slot, out, status);   // seam-aware
        if (r == SNK_LOOP_DONE) break;
    }
}

// grid: one workgroup per `lanes*waves` jobs.  dynamic LDS = 2 KiB LUT + 1904 B per chain.
__device__ __forceinline__ void snk_fast_kernel_body(const SnkTables &T, const SnkJob *jobs, uint32_t n_jobs,
                                                     uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    uint16_t *slot = (uint16_t *)snk_lds8;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t chains = lanes * waves;

    for (uint32_t t = tid; t < 512u; t += blockDim.x)
        ((uint32_t *)slot)[t] = ((const uint32_t *)T.lut_slot)[t];

    // chain c of the workgroup -> lane c / waves of wave c % waves  (spreads a
    // partially filled tail group over all waves)
    const uint32_t c = lane * waves + wave;
    const uint32_t j = blockIdx.x * chains + c;
    const bool active = lane < lanes && j < n_jobs;
    uint8_t *mine = snk_lds8 + SNK_FLUT_B + (size_t)(wave * lanes + (lane < lanes ? lane : 0u)) * SNK_FCHAIN_B;

    SnkJob job; job.xi = 0; job.yi = -1; job.out_idx = 0; job.snap = 0;
    if (active) job = jobs[j];

    // cooperative table initialisation from the prefix snapshot (or the all-zero start state)
    for (uint32_t l = 0; l < lanes; ++l) {
        const int a   = __shfl((int)active, (int)l);
        const int xi  = __shfl(job.xi, (int)l);
        const int snp = __shfl(job.snap, (int)l);
        if (!a) continue;
        uint8_t *dst = snk_lds8 + SNK_FLUT_B + (size_t)(wave * lanes + l) * SNK_FCHAIN_B;
        const uint32_t spos = T.snap_pos[xi];
        const bool use = (snp == 0) && (spos != 0u);
        const uint32_t *src = T.snap_fast + (size_t)xi * SNK_FSLOTS;
        for (uint32_t t = lane; t < SNK_FSLOTS / 2u; t += 64u) {
            uint32_t v = 0u;
            if (use) {
                const uint32_t a0 = src[2u * t], a1 = src[2u * t + 1u];
                const uint32_t lo = (a0 + 65536u >= spos) ? (a0 & 0xFFFFu) : 0u;   // previous block, else dead
                const uint32_t hi = (a1 + 65536u >= spos) ? (a1 & 0xFFFFu) : 0u;
                v = lo | (hi << 16);
            }
            ((uint32_t *)dst)[t] = v;
        }
        // no snapshot: stream start, every slot holds position 0 "written in this block"
        if (lane < SNK_FBMWORDS) ((uint32_t *)(dst + SNK_FSLOTS * 2u))[lane] = use ? 0u : 0xFFFFFFFFu;
    }
    __syncthreads();

    if (active) snk_fast_chain(T, job, mine, slot, out, status);
}

// phase B: ordered pairs (the dominant kernel of the bench)
__global__ void snk_fast_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body(T, jobs, n_jobs, lanes, out, status);
}

// phase A: single sequences + prefix snapshots at upload (same code, own symbol so that profiles
// keep the two phases apart)
__global__ void snk_fast_singles_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                        uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body(T, jobs, n_jobs, lanes, out, status);
}

// =========================================================================
//  generic byte kernel
// =========================================================================

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

template <bool LINKED>
__device__ __forceinline__ uint32_t snk_tget(const uint32_t *t32, uint32_t h)
{
    if (LINKED) return t32[h];
    return ((const uint16_t *)t32)[h];
}
template <bool LINKED>
__device__ __forceinline__ void snk_tput(uint32_t *t32, uint32_t h, uint32_t pos)
{
    if (LINKED) t32[h] = pos;
    else ((uint16_t *)t32)[h] = (uint16_t)pos;
}

// One block [pos, pos+blen) of the stream.  Returns the payload size (raw length
// when liblz4's limitedOutput compressor gives up).
template <bool LINKED>
__device__ __forceinline__ uint32_t snk_gen_block(const SnkGenSrc &s, uint32_t *tbl,
                                                  uint32_t pos, uint32_t blen,
                                                  uint64_t &guard, uint32_t *status)
{
    const uint32_t iend = pos + blen;
    if (blen < 13u) return blen;
    const uint32_t mfl1 = iend - 11u, mlimit = iend - 5u, olimit = blen - 1u;
    uint32_t cur, step = 1u, nb = 64u, anchor = pos, op = 0u;
    bool pending = false;

    {
        uint64_t w = snk_ld8(s, pos);
        snk_tput<LINKED>(tbl, LINKED ? snk_hash5(w) : snk_hash4(w), pos);
    }
    cur = pos + 1u;
    for (;;) {
        if (--guard == 0) { atomicOr(status, SNK_ST_ITERCAP); return blen; }
        const uint32_t next = cur + step;
        if (next > mfl1) break;
        if (pending) {
            uint64_t w2 = snk_ld8(s, cur - 2u);
            snk_tput<LINKED>(tbl, LINKED ? snk_hash5(w2) : snk_hash4(w2), cur - 2u);
        }
        const uint64_t wc = snk_ld8(s, cur);
        const uint32_t h = LINKED ? snk_hash5(wc) : snk_hash4(wc);
        uint32_t cand = snk_tget<LINKED>(tbl, h);
        snk_tput<LINKED>(tbl, h, cur);
        { uint32_t s2 = nb >> 6; step = s2 ? s2 : 1u; nb++; }
        const uint64_t wd = snk_ld8(s, cand);
        const bool near = LINKED ? (cand + SNK_MAXDIST >= cur) : true;
        if (near && (uint32_t)wc == (uint32_t)wd) {
            uint32_t ip = cur;
            while (ip > anchor && cand > 0u && snk_byte_at(s, ip - 1u) == snk_byte_at(s, cand - 1u)) { ip--; cand--; }
            const uint32_t lit = ip - anchor;
            op += 1u;
            if (op + lit + 8u + lit / 255u > olimit) return blen;
            op += lit + snk_lit_ext(lit) + 2u;
            // forward count from ip+4 / cand+4, capped at mlimit
            uint32_t a = ip + 4u, b = cand + 4u;
            while (a < mlimit) {
                uint64_t d = snk_ld8(s, a) ^ snk_ld8(s, b);
                if (d) { a += (uint32_t)__builtin_ctzll(d) >> 3; break; }
                a += 8u; b += 8u;
            }
            if (a > mlimit) a = mlimit;
            const uint32_t mc = a - (ip + 4u);
            if (op + 6u + (mc + 240u) / 255u > olimit) return blen;
            if (mc >= 15u) op += (mc - 15u) / 255u + 1u;
            anchor = a;
            cur = a; step = 1u; nb = 63u; pending = true;
            if (a >= mfl1) break;
        } else {
            cur = next; pending = false;
        }
    }
    {
        const uint32_t run = iend - anchor;
        if (op + run + 1u + (run + 240u) / 255u > olimit) return blen;
        return op + 1u + snk_lit_ext(run) + run;
    }
}

__device__ __forceinline__ void snk_gen_chain(const SnkTables &T, const SnkJob job,
                                              uint32_t *tbl, uint32_t *out, uint32_t *status)
{
    SnkGenSrc s;
    const uint32_t lx = T.len[job.xi];
    const uint32_t ly = job.yi >= 0 ? T.len[job.yi] : 0u;
    const uint32_t n = lx + ly;
    s.xb = T.bytes[job.xi];
    s.yb = job.yi >= 0 ? T.bytes[job.yi] : T.zero_pad + SNK_PAD;
    s.lx = lx;
    uint64_t guard = 2ull * n + 4096ull;

    if (n == 0u) { out[job.out_idx] = T.header_bytes + 4u; return; }
    if (n <= SNK_BLOCK) {                         // one independent block, one-shot compressor
        uint32_t payload = snk_gen_block<false>(s, tbl, 0u, n, guard, status);
        out[job.out_idx] = T.header_bytes + 4u + payload + 4u;
        return;
    }
    uint32_t pos, total;
    const uint32_t spos = T.snap_pos[job.xi];
    if (job.snap == 0 && spos != 0u) { pos = spos; total = T.snap_out[job.xi]; }
    else                             { pos = 0u;   total = T.header_bytes; }
    while (pos < n) {
        const uint32_t blen = n - pos < SNK_BLOCK ? n - pos : SNK_BLOCK;
        total += 4u + snk_gen_block<true>(s, tbl, pos, blen, guard, status);
        pos += blen;
        if (job.snap != 0 && pos == spos) {
            uint32_t *dst = T.snap_gen + (size_t)job.xi * 4096u;
            for (uint32_t t = 0; t < 4096u; ++t) dst[t] = tbl[t];
            T.snap_out[job.xi] = total;
        }
    }
    out[job.out_idx] = total + 4u;
}

// grid: one 64-thread workgroup per `chains` jobs; dynamic LDS = 16 KiB per chain.
__global__ void snk_generic_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                   uint32_t chains, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t snk_lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t j = blockIdx.x * chains + lane;
    const bool active = lane < chains && j < n_jobs;
    uint32_t *tbl = snk_lds + (size_t)(lane < chains ? lane : 0u) * 4096u;

    SnkJob job; job.xi = 0; job.yi = -1; job.out_idx = 0; job.snap = 0;
    if (active) job = jobs[j];

    for (uint32_t l = 0; l < chains; ++l) {
        const int a   = __shfl((int)active, (int)l);
        const int xi  = __shfl(job.xi, (int)l);
        const int yi  = __shfl(job.yi, (int)l);
        const int snp = __shfl(job.snap, (int)l);
        if (!a) continue;
        uint32_t *dst = snk_lds + (size_t)l * 4096u;
        const uint32_t n = T.len[xi] + (yi >= 0 ? T.len[yi] : 0u);
        const bool use = (snp == 0) && (T.snap_pos[xi] != 0u) && n > SNK_BLOCK;
        const uint32_t *src = T.snap_gen + (size_t)xi * 4096u;
        for (uint32_t t = lane; t < 4096u; t += 64u) dst[t] = use ? src[t] : 0u;
    }
    __syncthreads();

    if (active) snk_gen_chain(T, job, tbl, out, status);
}

// =========================================================================
//  byte kernel, linked mode (n > 64 KiB), any alphabet: N runs, lower case, protein ...
// =========================================================================
//
// Same execution model as the 2-bit kernel (one lane = one chain, flat probe loop, side exits
// that are wave-uniform, cursor-side register reservoir) on ASCII data.  The table is liblz4's
// full 4096-slot table, kept as 16-bit block offsets + "written this block" bitmap exactly like
// the 2-bit kernel: 8712 B per chain -> 18 chains per CU (the u32 table of snk_generic_kernel
// allows 8).  The slot is liblz4's 12-bit hash of 5 bytes, computed arithmetically.
// Two table geometries (template parameter COMPACT):
//   full    : slot = hash (4096 slots + 1 dummy), 8712 B per chain -> 18 chains per CU.
// ONESHOT = liblz4's one-shot mode for inputs <= 64 KiB (a single independent block): the slot is
// the 13-bit hash of 4 bytes and there is no distance limit -- with one block every entry is
// "current", so the same table logic applies unchanged.
//   compact : when the 5-byte hashes that occur in ANY resident sequence number <= 1024 (upper-case
//             ACGT with N runs and a few IUPAC codes: typically 900-1000) or <= 2048 (soft-masked
//             genomes), a shared LUT renames them to 0..CAP-1 (exact: it is a renaming).  Only the <= 4 five-byte strings that span the
//             x/y seam of a pair can hash outside that set; they get the chain-private slots
//             CAP..CAP+3.  2196 / 4380 B per chain (+ 8 KiB LUT per workgroup) -> 70 / 35 chains per CU.
template <int CAP, bool ONESHOT> struct SnkBT {                   // CAP: 0 = full table, 1024 / 2048 = compact capacity
    static constexpr bool     COMPACT = CAP != 0;
    static constexpr uint32_t HASHES  = ONESHOT ? 8192u : 4096u;        // liblz4: 13-bit hash of 4 bytes / 12-bit of 5
    static constexpr uint32_t SEAM0   = (uint32_t)CAP;                  // first of the seam-private slots
    static constexpr uint32_t SLOTS   = COMPACT ? (uint32_t)CAP + 4u : HASHES;   // real slots
    static constexpr uint32_t DUMMY   = SLOTS;                          // absorbs the put of "nothing owed"
    static constexpr uint32_t TBL_B   = ((SLOTS + 1u + 3u) / 4u) * 8u;  // u16 entries, rounded to 8 bytes
    static constexpr uint32_t BMWORDS = (SLOTS + 1u + 31u) / 32u;
    static constexpr uint32_t CHAIN_B = TBL_B + BMWORDS * 4u;           // 2196 (1024) / 4372 (2048) / 8716 (4096) / 17420 (8192)
    static constexpr uint32_t LUT_B   = COMPACT ? HASHES * 2u : 0u;     // hash -> slot LUT at LDS offset 0
    static constexpr uint32_t KBYTES  = ONESHOT ? 4u : 5u;              // bytes hashed per position
};
#define SNK_BC_NOSLOT   0xFFFFu

struct SnkByteSrc {
    snk_g8 *arena;            // wave-uniform base of the ASCII arena
    uint32_t xoff, yoff;      // byte offsets of the two sequences
    uint32_t lx;
};

__device__ __forceinline__ uint32_t snk_hash5_parts(uint32_t a_lo, uint32_t a_hi)
{
    // ((seq << 24) * 889523592379) >> 52 with (seq << 24) = a_hi:a_lo
    const uint64_t a = ((uint64_t)a_hi << 32) | a_lo;
    return (uint32_t)((a * 889523592379ull) >> 52);
}

// 8 bytes of the concatenation starting at p, seam aware (slow paths only)
__device__ __forceinline__ uint64_t snk_bld8(const SnkByteSrc &s, uint32_t p)
{
    if (p + 8u <= s.lx) return snk_ld8g(s.arena + (size_t)(s.xoff + p));
    if (p >= s.lx) return snk_ld8g(s.arena + (size_t)(s.yoff + (p - s.lx)));
    const uint32_t k = s.lx - p;                               // 1..7 bytes from x, rest from y
    const uint64_t xv = snk_ld8g(s.arena + (size_t)(s.xoff + p));      // zero beyond lx (padding)
    const uint64_t yv = snk_ld8g(s.arena + (size_t)s.yoff);
    return xv | (yv << (8u * k));
}
__device__ __forceinline__ uint32_t snk_bbyte(const SnkByteSrc &s, uint32_t p)
{
    return p < s.lx ? s.arena[(size_t)(s.xoff + p)] : s.arena[(size_t)(s.yoff + (p - s.lx))];
}

struct __attribute__((packed)) SnkU96 { uint32_t a, b, c; };
struct SnkW12 { uint32_t a, b, c; };               // 12 bytes [p-4, p+8): a = p-4..p-1, b = p..p+3, c = p+4..p+7

// Candidate window for the tight loop: the 12 bytes [p-4, p+8) of the concatenation.  The source
// (x or y) is chosen by select; a window that straddles the seam is assembled by the slow loader
// (wave-uniform branch, taken only within 12 bytes of the seam).
__device__ __forceinline__ SnkW12 snk_bfetch12(const SnkByteSrc &s, uint32_t p)
{
    const int32_t q0 = (int32_t)p - 4;
    const bool iny = (q0 >= (int32_t)s.lx);
    const bool straddle = !iny & (p + 8u > s.lx);
    const uint32_t off = iny ? s.yoff + (uint32_t)(q0 - (int32_t)s.lx) : (uint32_t)((int32_t)s.xoff + q0);
    const __attribute__((address_space(1))) SnkU96 *vp = (const __attribute__((address_space(1))) SnkU96 *)(s.arena + (size_t)off);
    SnkW12 r; r.a = vp->a; r.b = vp->b; r.c = vp->c;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(straddle) != 0ull, 0)) {
        if (straddle) {
            // byte by byte: the window may start before position 0 (candidate < 4 with a prefix
            // shorter than 12 bytes); those bytes are never used (catch-up is capped by cand)
            uint32_t v[3] = { 0u, 0u, 0u };
            for (int32_t k = 0; k < 12; ++k) {
                const int32_t pos = q0 + k;
                const uint32_t byte = pos < 0 ? 0u : snk_bbyte(s, (uint32_t)pos);
                v[k >> 2] |= byte << (8 * (k & 3));
            }
            r.a = v[0]; r.b = v[1]; r.c = v[2];
        }
    }
    return r;
}

// Cursor-side reservoir: 24 bytes [rb, rb+24) of ONE source in registers plus the next 8 in
// flight.  A probe at cur needs the bytes [cur-4, cur+8): offset o = cur-4-rb must be 0..7.
struct SnkBWin {
    uint32_t soff, org, rb, lim;       // lim = largest cursor this source can serve (0 = unusable)
    uint32_t r0, r1, r2, r3, r4, r5, nx0, nx1;
};

__device__ __forceinline__ void snk_bwin_init(SnkBWin &w, snk_g8 *arena, uint32_t soff, uint32_t org,
                                              uint32_t lim, uint32_t cur)
{
    w.soff = soff; w.org = org; w.lim = lim;
    w.rb = org + ((cur - 4u - org) & ~3u);
    snk_g8 *p = arena + (size_t)(soff + (w.rb - org));
    w.r0 = snk_ld4g(p); w.r1 = snk_ld4g(p + 4); w.r2 = snk_ld4g(p + 8); w.r3 = snk_ld4g(p + 12);
    w.r4 = snk_ld4g(p + 16); w.r5 = snk_ld4g(p + 20); w.nx0 = snk_ld4g(p + 24); w.nx1 = snk_ld4g(p + 28);
}

struct SnkByteLane {
    SnkByteSrc s;
    uint32_t n, spos;
    int32_t xi, snap;
    uint32_t out_idx;
    uint32_t pos, total, iend, blen, blocks_left;
    bool first, in_block;
    uint32_t cur, step, nb, anchor, op;
    uint32_t mfl1, mlimit, olimit, base;
    uint32_t endcode;
    bool pending;
    SnkBWin w;
};

// Data of one probe taken from the reservoir at byte offset o (0..7): the 12-byte compare window
// and the two table slots (5 bytes at cur and at cur-2).
struct SnkBProbeData { SnkW12 w; uint32_t s1, s2; };

__device__ __forceinline__ uint32_t snk_hash4_u32(uint32_t v) { return (v * 2654435761u) >> 19; }

// hash -> table slot (compact: through the LUT at LDS address 0; the kernel has no static LDS)
template <bool COMPACT>
__device__ __forceinline__ uint32_t snk_bslot(uint32_t h)
{
    if (!COMPACT) return h;
    const __attribute__((address_space(3))) uint16_t *const lut = (const __attribute__((address_space(3))) uint16_t *)0;
    return lut[h];
}

template <int CAP, bool ONESHOT>
__device__ __forceinline__ SnkBProbeData snk_bextract(const SnkBWin &w, uint32_t o)
{
    const bool hi = (o & 4u) != 0u;
    const uint32_t sh = (o & 3u) * 8u;
    const uint32_t a0 = hi ? w.r1 : w.r0, a1 = hi ? w.r2 : w.r1, a2 = hi ? w.r3 : w.r2, a3 = hi ? w.r4 : w.r3;
    SnkBProbeData d;
    d.w.a = __builtin_amdgcn_alignbit(a1, a0, sh);       // bytes cur-4 .. cur-1
    d.w.b = __builtin_amdgcn_alignbit(a2, a1, sh);       // bytes cur   .. cur+3
    d.w.c = __builtin_amdgcn_alignbit(a3, a2, sh);       // bytes cur+4 .. cur+7
    // 5 bytes at cur-2 = window bytes 2..6 ; 5 bytes at cur = window bytes 4..8
    if (ONESHOT) {      // 4 bytes at cur-2 = window bytes 2..5 ; 4 bytes at cur = window bytes 4..7
        d.s2 = snk_bslot<(CAP != 0)>(snk_hash4_u32(__builtin_amdgcn_alignbit(d.w.b, d.w.a, 16)));
        d.s1 = snk_bslot<(CAP != 0)>(snk_hash4_u32(d.w.b));
    } else {
        d.s2 = snk_bslot<(CAP != 0)>(snk_hash5_parts((d.w.a << 8) & 0xFF000000u, __builtin_amdgcn_alignbit(d.w.b, d.w.a, 24)));
        d.s1 = snk_bslot<(CAP != 0)>(snk_hash5_parts(d.w.b << 24, __builtin_amdgcn_alignbit(d.w.c, d.w.b, 8)));
    }
    return d;
}

// Slot of the 5 bytes at stream position p, for the slow paths (direct loads, seam aware).  In compact
// mode a hash outside the resident set can only belong to a string spanning the seam (p in
// [lx-4, lx-1]); equal seam hashes share one private slot, as they would share liblz4's.
template <int CAP, bool ONESHOT>
__device__ __forceinline__ uint32_t snk_bslot_slow(const SnkByteSrc &s, uint32_t p)
{
    const uint64_t w0 = snk_bld8(s, p);
    const uint32_t h = ONESHOT ? snk_hash4_u32((uint32_t)w0) : snk_hash5_parts((uint32_t)w0 << 24, (uint32_t)(w0 >> 8));
    if (CAP == 0) return h;
    const uint32_t id = snk_bslot<true>(h);
    if (__builtin_expect(id != SNK_BC_NOSLOT, 1)) return id;
    const int32_t j0 = (int32_t)s.lx - (int32_t)(SnkBT<CAP, ONESHOT>::KBYTES - 1u);   // first seam-spanning position
    int32_t q = j0 < 0 ? 0 : j0;
    for (; q < (int32_t)p; ++q) {
        const uint64_t wq = snk_bld8(s, (uint32_t)q);
        const uint32_t hq = ONESHOT ? snk_hash4_u32((uint32_t)wq) : snk_hash5_parts((uint32_t)wq << 24, (uint32_t)(wq >> 8));
        if (hq == h) break;
    }
    return SnkBT<CAP, ONESHOT>::SEAM0 + (uint32_t)(q - j0);
}

template <int CAP, bool ONESHOT>
__device__ __forceinline__ bool snk_bytes_block_step(SnkByteLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                                     uint32_t *out, uint32_t *status)
{
    if (L.in_block) {
        uint32_t payload = L.blen;
        if (L.endcode != 2u) {
            const uint32_t run = L.iend - L.anchor;
            if (L.op + run + 1u + (run + 240u) / 255u <= L.olimit)
                payload = L.op + 1u + snk_lit_ext(run) + run;
        }
        L.total += 4u + payload;
        L.pos = L.iend;
        L.in_block = false;
        L.endcode = 0u;
    }
    for (;;) {
        if (L.snap != 0 && L.pos == L.spos && L.spos != 0u) {
            // prefix snapshot, always in liblz4's hash-indexed form (absolute positions, 0 = too far)
            uint32_t *dst = T.snap_gen + (size_t)L.xi * 4096u;
            for (uint32_t h = 0; h < 4096u; ++h) {        // (never taken in one-shot mode: no snapshots)
                const uint32_t t = snk_bslot<(CAP != 0)>(h);
                uint32_t v = 0u;
                if (CAP == 0 || t != SNK_BC_NOSLOT) {
                    if ((bm[t >> 5] >> (t & 31u)) & 1u) v = L.pos - 65536u + tbl[t];
                }
                dst[h] = v;
            }
            T.snap_out[L.xi] = L.total;
        }
        if (L.pos >= L.n) { out[L.out_idx] = L.total + 4u; return true; }
        if (L.blocks_left-- == 0u) { atomicOr(status, SNK_ST_ITERCAP); return true; }
        L.blen = L.n - L.pos < SNK_BLOCK ? L.n - L.pos : SNK_BLOCK;
        L.iend = L.pos + L.blen;
        if (L.blen < 13u) { L.total += 4u + L.blen; L.pos = L.iend; continue; }
        if (!L.first) {
            for (uint32_t wi = 0; wi < (SnkBT<CAP, ONESHOT>::SLOTS + 31u) / 32u; ++wi) {
                uint32_t z = ~bm[wi];
                while (z) {
                    const uint32_t b = (uint32_t)__builtin_ctz(z);
                    tbl[wi * 32u + b] = 0;
                    z &= z - 1u;
                }
                bm[wi] = 0u;
            }
        }
        L.first = false;
        L.base = L.pos;
        L.mfl1 = L.iend - 11u; L.mlimit = L.iend - 5u; L.olimit = L.blen - 1u;
        {
            const uint32_t s0 = snk_bslot_slow<CAP, ONESHOT>(L.s, L.pos);
            tbl[s0] = 0;
            atomicOr(&bm[s0 >> 5], 1u << (s0 & 31u));
        }
        L.cur = L.pos + 1u; L.step = 1u; L.nb = 64u; L.anchor = L.pos; L.op = 0u;
        L.pending = false; L.in_block = true;
        return false;
    }
}

// liblz4's exact handling of a match found at cur with candidate cand (slow, general).
__device__ __forceinline__ void snk_bytes_match_slow(SnkByteLane &L, uint32_t cur, uint32_t cand,
                                                     uint32_t anchor0, uint32_t op0)
{
    const SnkByteSrc &s = L.s;
    uint32_t ip = cur;
    while (ip > anchor0 && cand > 0u && snk_bbyte(s, ip - 1u) == snk_bbyte(s, cand - 1u)) { ip--; cand--; }
    const uint32_t lit = ip - anchor0;
    uint32_t a = ip + 4u, b = cand + 4u;
    while (a < L.mlimit) {
        const uint64_t d = snk_bld8(s, a) ^ snk_bld8(s, b);
        if (d) { a += (uint32_t)__builtin_ctzll(d) >> 3; break; }
        a += 8u; b += 8u;
    }
    if (a > L.mlimit) a = L.mlimit;
    const uint32_t mc = a - (ip + 4u);
    uint32_t op = op0 + 1u;
    bool bail = op + lit + 8u + lit / 255u > L.olimit;
    if (!bail) {
        op += lit + snk_lit_ext(lit) + 2u;
        bail = op + 6u + (mc + 240u) / 255u > L.olimit;
        if (mc >= 15u) op += (mc - 15u) / 255u + 1u;
    }
    if (bail) { L.endcode = 2u; L.mfl1 = 0u; L.step = 1u; L.cur = cur; L.anchor = anchor0; L.op = op0; return; }
    L.op = op;
    L.anchor = a;
    L.cur = a; L.step = 1u; L.nb = 63u; L.pending = true;
    if (a >= L.mfl1) { L.endcode = 1u; L.mfl1 = 0u; }
}

// table probe shared by the slow and the tight paths: returns candidate + validity, performs the puts
template <int CAP, bool ONESHOT>
__device__ __forceinline__ void snk_bytes_table(const SnkByteLane &L, uint16_t *tbl, uint32_t *bm, uint32_t cur,
                                                uint32_t s1, uint32_t s2, uint32_t &cand, bool &valid)
{
    s2 = L.pending ? s2 : SnkBT<CAP, ONESHOT>::DUMMY;
    const uint32_t e = tbl[s1];
    const uint32_t bw = bm[s1 >> 5];
    const uint32_t c = cur - L.base;
    const uint32_t bit1 = 1u << (s1 & 31u);
    tbl[s2] = (uint16_t)(c - 2u);
    atomicOr(&bm[s2 >> 5], 1u << (s2 & 31u));
    tbl[s1] = (uint16_t)c;
    atomicOr(&bm[s1 >> 5], bit1);
    const bool iscur = (bw & bit1) != 0u;
    cand = L.base + e - (iscur ? 0u : 65536u);
    valid = iscur | (e > c);
    const bool same = (s2 == s1);
    cand = same ? cur - 2u : cand;
    valid |= same;
    cand = valid ? cand : cur;
}

// One fully general probe with direct loads (stream start, seam, after long jumps).
template <int CAP, bool ONESHOT>
__device__ __forceinline__ bool snk_bytes_iter_slow(SnkByteLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                                    uint32_t *out, uint32_t *status)
{
    const uint32_t cur = L.cur, next = cur + L.step;
    if (next > L.mfl1) return snk_bytes_block_step<CAP, ONESHOT>(L, T, tbl, bm, out, status);
    const uint64_t wc = snk_bld8(L.s, cur);
    const uint32_t s1 = snk_bslot_slow<CAP, ONESHOT>(L.s, cur);
    // the put of cur-2 is owed only after a match, which ends at least 5 positions into the block
    const uint32_t s2 = L.pending ? snk_bslot_slow<CAP, ONESHOT>(L.s, cur - 2u) : SnkBT<CAP, ONESHOT>::DUMMY;
    uint32_t cand; bool valid;
    snk_bytes_table<CAP, ONESHOT>(L, tbl, bm, cur, s1, s2, cand, valid);
    const uint32_t s3 = L.nb >> 6;
    const uint64_t wd = snk_bld8(L.s, cand);
    if (valid && (uint32_t)wc == (uint32_t)wd) {
        snk_bytes_match_slow(L, cur, cand, L.anchor, L.op);
    } else {
        L.cur = next; L.step = s3 ? s3 : 1u; L.nb++; L.pending = false;
    }
    // seat the reservoir for the tight loop when the new cursor allows it
    const uint32_t nc = L.cur;
    if (nc >= L.s.lx + 4u)                         snk_bwin_init(L.w, L.s.arena, L.s.yoff, L.s.lx, 0xFFFFFFFFu, nc);
    else if (nc >= 4u && nc + 8u <= L.s.lx)        snk_bwin_init(L.w, L.s.arena, L.s.xoff, 0u, L.s.lx - 8u, nc);
    else                                           L.w.lim = 0u;
    return false;
}

// Tight loop.  Invariant at the head: (nx0, nx1) hold the bytes [rb+24, rb+32).
template <int CAP, bool ONESHOT>
__device__ __forceinline__ void snk_bytes_loop(SnkByteLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                               uint32_t *out, uint32_t *status)
{
    SnkBWin &w = L.w;
    snk_g8 *const arena = L.s.arena;
    for (;;) {
        uint32_t cur, next, o;
        for (;;) {
            cur = L.cur;
            next = cur + L.step;
            o = cur - 4u - w.rb;
            const bool pre = (next > L.mfl1) | (o > 7u) | (cur > w.lim);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pre) == 0ull, 1)) break;
            if (pre && snk_bytes_iter_slow<CAP, ONESHOT>(L, T, tbl, bm, out, status)) return;
        }
        SnkBProbeData d = snk_bextract<CAP, ONESHOT>(w, o);
        const uint32_t olim6 = L.olimit - 6u;

        for (;;) {
            uint32_t cand; bool valid;
            snk_bytes_table<CAP, ONESHOT>(L, tbl, bm, cur, d.s1, d.s2, cand, valid);

            __builtin_amdgcn_sched_barrier(0);
            snk_g8 *nxp = arena + (size_t)(w.soff + (w.rb + 24u - w.org));
            const SnkW12 wd = snk_bfetch12(L.s, cand);
            const uint64_t nxv = snk_ld8g(nxp);
            __builtin_amdgcn_sched_barrier(0);
            w.nx0 = (uint32_t)nxv; w.nx1 = (uint32_t)(nxv >> 32);

            const uint32_t x0 = d.w.a ^ wd.a, x1 = d.w.b ^ wd.b, x2 = d.w.c ^ wd.c;
            // equal bytes forward from cur (0..8) and backward before cur (0..4)
            uint32_t fh = (uint32_t)__builtin_ctz(x2 | 0x80000000u) >> 3;          // 0..3, 4 when x2 == 0 is handled below
            fh = x2 ? fh : 4u;
            const uint32_t f = x1 ? ((uint32_t)__builtin_ctz(x1) >> 3) : 4u + fh;
            const uint32_t eq = x0 ? ((uint32_t)__builtin_clz(x0) >> 3) : 4u;
            const bool m = valid & (x1 == 0u);
            uint32_t e2 = cur + f;
            e2 = e2 < L.mlimit ? e2 : L.mlimit;
            const uint32_t s3 = L.nb >> 6;
            const uint32_t nstep = m ? 1u : (s3 ? s3 : 1u);
            const uint32_t ncur = m ? e2 : next;
            const uint32_t nnext = ncur + nstep;

            // ---- next probe's data from the reservoir ----
            uint32_t no = ncur - 4u - w.rb;
            const bool sl = (no - 8u) < 8u;                   // slide by 8 bytes
            const uint32_t r0n = sl ? w.r2 : w.r0, r1n = sl ? w.r3 : w.r1, r2n = sl ? w.r4 : w.r2;
            const uint32_t r3n = sl ? w.r5 : w.r3, r4n = sl ? w.nx0 : w.r4, r5n = sl ? w.nx1 : w.r5;
            no -= sl ? 8u : 0u;
            w.r0 = r0n; w.r1 = r1n; w.r2 = r2n; w.r3 = r3n; w.r4 = r4n; w.r5 = r5n; w.rb += sl ? 8u : 0u;
            const SnkBProbeData nd = snk_bextract<CAP, ONESHOT>(w, no & 7u);
            if (CAP != 0) __builtin_amdgcn_sched_barrier(0);   // keep the LUT reads in front of the bookkeeping

            // ---- bookkeeping of this probe ----
            const uint32_t anchor0 = L.anchor, op0 = L.op;
            uint32_t lit = cur - anchor0;
            uint32_t b = eq < lit ? eq : lit;
            b = b < cand ? b : cand;
            lit -= b;
            const uint32_t opn = op0 + lit + 3u;
            // rare: catch-up reaches 4, match reaches 8, literal run needs extension bytes, budget
            // (match code = f + b - 4 <= 7 never needs extension bytes here)
            const bool rare = m & ((b == 4u) | (f == 8u) | (lit >= 15u) | (opn > olim6));
            const bool pre = (nnext > L.mfl1) | (no > 7u) | (ncur > w.lim);
            L.op = m ? opn : op0;
            L.anchor = m ? e2 : anchor0;
            L.step = nstep;
            L.nb = m ? 63u : L.nb + 1u;
            L.cur = ncur;
            L.pending = m;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare | pre) != 0ull, 0)) {
                if (rare) snk_bytes_match_slow(L, cur, cand, anchor0, op0);
                const uint64_t rf = snk_ld8g(arena + (size_t)(w.soff + (w.rb + 24u - w.org)));
                w.nx0 = (uint32_t)rf; w.nx1 = (uint32_t)(rf >> 32);
                break;
            }
            cur = ncur; next = nnext; d = nd;
        }
    }
}

// grid: one workgroup per `lanes*waves` jobs; dynamic LDS = LUT_B + CHAIN_B per chain.
template <int CAP, bool ONESHOT>
__device__ __forceinline__ void snk_bytes_kernel_body(const SnkTables &T, const SnkJob *jobs, uint32_t n_jobs,
                                                      uint32_t lanes, uint32_t *out, uint32_t *status)
{
    typedef SnkBT<CAP, ONESHOT> G;
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t chains = lanes * waves;
    const uint32_t c = lane * waves + wave;
    const uint32_t j = blockIdx.x * chains + c;
    const bool active = lane < lanes && j < n_jobs;
    uint8_t *mine = snk_lds8 + G::LUT_B + (size_t)(wave * lanes + (lane < lanes ? lane : 0u)) * G::CHAIN_B;

    if (CAP != 0) {
        const uint32_t *lsrc = (const uint32_t *)(ONESHOT ? T.lut_h2c4 : T.lut_h2c);
        for (uint32_t t = tid; t < G::LUT_B / 4u; t += blockDim.x)
            ((uint32_t *)snk_lds8)[t] = lsrc[t];
        __syncthreads();
    }
    SnkJob job; job.xi = 0; job.yi = -1; job.out_idx = 0; job.snap = 0;
    if (active) job = jobs[j];

    for (uint32_t l = 0; l < lanes; ++l) {
        const int a   = __shfl((int)active, (int)l);
        const int xi  = __shfl(job.xi, (int)l);
        const int snp = __shfl(job.snap, (int)l);
        if (!a) continue;
        uint8_t *dst = snk_lds8 + G::LUT_B + (size_t)(wave * lanes + l) * G::CHAIN_B;
        const uint32_t spos = T.snap_pos[xi];
        const bool use = !ONESHOT && (snp == 0) && (spos != 0u);
        const uint32_t *src = T.snap_gen + (size_t)xi * 4096u;
        if (CAP == 0) {
            for (uint32_t t = lane; t < G::TBL_B / 4u; t += 64u) {
                uint32_t v = 0u;
                if (use && t < 2048u) {
                    const uint32_t a0 = src[2u * t], a1 = src[2u * t + 1u];
                    const uint32_t lo = (a0 + 65536u >= spos) ? (a0 & 0xFFFFu) : 0u;
                    const uint32_t hi = (a1 + 65536u >= spos) ? (a1 & 0xFFFFu) : 0u;
                    v = lo | (hi << 16);
                }
                ((uint32_t *)dst)[t] = v;
            }
        } else {
            for (uint32_t t = lane; t < G::TBL_B / 4u; t += 64u) ((uint32_t *)dst)[t] = 0u;
            if (use) {
                // scatter liblz4's hash-indexed snapshot into the renamed slots (LDS ops of a wave are
                // executed in issue order, so the zero fill above lands first)
                for (uint32_t h = lane; h < 4096u; h += 64u) {
                    const uint32_t id = snk_bslot<true>(h);
                    const uint32_t a0 = src[h];
                    if (id != SNK_BC_NOSLOT) ((uint16_t *)dst)[id] = (uint16_t)((a0 + 65536u >= spos) ? (a0 & 0xFFFFu) : 0u);
                }
            }
        }
        for (uint32_t t = lane; t < G::BMWORDS; t += 64u)
            ((uint32_t *)(dst + G::TBL_B))[t] = use ? 0u : 0xFFFFFFFFu;
    }
    __syncthreads();
    if (!active) return;

    uint16_t *tbl = (uint16_t *)mine;
    uint32_t *bm = (uint32_t *)(mine + G::TBL_B);
    SnkByteLane L;
    const uint32_t lx = T.len[job.xi];
    const uint32_t ly = job.yi >= 0 ? T.len[job.yi] : 0u;
    L.s.arena = (snk_g8 *)T.bytes_arena;
    L.s.xoff = T.bytes_off[job.xi];
    L.s.yoff = job.yi >= 0 ? T.bytes_off[job.yi] : 16u;       // zero region at the arena start
    L.s.lx = lx;
    L.n = lx + ly;
    L.spos = T.snap_pos[job.xi];
    L.xi = job.xi; L.snap = job.snap; L.out_idx = job.out_idx;
    if (!ONESHOT && job.snap == 0 && L.spos != 0u) { L.pos = L.spos; L.total = T.snap_out[job.xi]; }
    else                                           { L.pos = 0u;     L.total = T.header_bytes; }
    if (ONESHOT) L.snap = 0;
    L.blocks_left = (L.n >> 16) + 4u;
    L.iend = 0; L.blen = 0; L.first = true; L.in_block = false;
    L.cur = 0; L.step = 1; L.nb = 64; L.anchor = 0; L.op = 0;
    L.mfl1 = 0; L.mlimit = 0; L.olimit = 0; L.base = L.pos; L.endcode = 0;
    L.pending = false;
    L.w.soff = L.s.xoff; L.w.org = 0u; L.w.rb = 0u; L.w.lim = 0u;
    L.w.r0 = L.w.r1 = L.w.r2 = L.w.r3 = L.w.r4 = L.w.r5 = L.w.nx0 = L.w.nx1 = 0u;
    snk_bytes_loop<CAP, ONESHOT>(L, T, tbl, bm, out, status);
}

__global__ void snk_bytes_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                 uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<0, false>(T, jobs, n_jobs, lanes, out, status);
}

__global__ void snk_bytes_compact_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                         uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<1024, false>(T, jobs, n_jobs, lanes, out, status);
}

__global__ void snk_bytes_compact2k_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                           uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<2048, false>(T, jobs, n_jobs, lanes, out, status);
}

// one-shot mode (n <= 64 KiB): full 8192-slot table, and the compact form
__global__ void snk_oneshot_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                   uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<0, true>(T, jobs, n_jobs, lanes, out, status);
}

__global__ void snk_oneshot_compact_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                           uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<1024, true>(T, jobs, n_jobs, lanes, out, status);
}

// =========================================================================
//  ingest kernels
// =========================================================================

// flags[g] bit0 is cleared when a byte outside {A,C,G,T} is seen.
__global__ void snk_classify_kernel(const uint8_t *bytes, uint64_t n, uint32_t *flag)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < n; i += stride) {
        uint8_t c = bytes[i];
        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, ~1u);
}

// 2-bit pack: code = (c >> 1) & 3  (A=0, C=1, T=2, G=3); one output byte per thread.
__global__ void snk_pack_kernel(const uint8_t *bytes, uint64_t n, uint8_t *packed)
{
    uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nbytes = (n + 3) >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; o < nbytes; o += stride) {
        uint32_t v = 0;
        for (uint32_t b = 0; b < 4u; ++b) {
            uint64_t i = o * 4 + b;
            uint32_t code = i < n ? ((bytes[i] >> 1) & 3u) : 0u;
            v |= code << (2u * b);
        }
        packed[o] = (uint8_t)v;
    }
}

// Which of liblz4's 4096 hash values occur inside one sequence (5 bytes at every position p <= n-5).
// One 4096-bit set per launch target, OR-ed into `set` (128 words).
__global__ void snk_hashset_kernel(const uint8_t *bytes, uint64_t n, uint32_t *set)
{
    __shared__ uint32_t local[128];
    for (uint32_t t = threadIdx.x; t < 128u; t += blockDim.x) local[t] = 0u;
    __syncthreads();
    if (n >= 5) {
        const uint64_t last = n - 5;
        uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (; p <= last; p += stride) {
            uint64_t v = 0;
            for (uint32_t b = 0; b < 5u; ++b) v |= (uint64_t)bytes[p + b] << (8u * b);
            const uint32_t h = (uint32_t)(((v << 24) * 889523592379ull) >> 52);
            atomicOr(&local[h >> 5], 1u << (h & 31u));
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 128u; t += blockDim.x)
        if (local[t]) atomicOr(&set[t], local[t]);
}

// Same for the one-shot hash (13 bits of 4 bytes, positions p <= n-4); `set` has 256 words.
__global__ void snk_hashset4_kernel(const uint8_t *bytes, uint64_t n, uint32_t *set)
{
    __shared__ uint32_t local[256];
    for (uint32_t t = threadIdx.x; t < 256u; t += blockDim.x) local[t] = 0u;
    __syncthreads();
    if (n >= 4) {
        const uint64_t last = n - 4;
        uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (; p <= last; p += stride) {
            uint32_t v = 0;
            for (uint32_t b = 0; b < 4u; ++b) v |= (uint32_t)bytes[p + b] << (8u * b);
            const uint32_t h = (v * 2654435761u) >> 19;
            atomicOr(&local[h >> 5], 1u << (h & 31u));
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 256u; t += blockDim.x)
        if (local[t]) atomicOr(&set[t], local[t]);
}

// slot-indexed snapshot -> hash-indexed snapshot (for ACGT prefix + non-ACGT suffix pairs)
__global__ void snk_snap_convert_kernel(const uint32_t *snap_fast, uint32_t *snap_gen,
                                        const uint32_t *lut_hash, const uint16_t *lut_slot,
                                        const uint32_t *seq_ids, uint32_t n_ids)
{
    const uint32_t g = seq_ids[blockIdx.x];
    (void)n_ids;
    uint32_t *dst = snap_gen + (size_t)g * 4096u;
    const uint32_t *src = snap_fast + (size_t)g * SNK_FSLOTS;
    for (uint32_t t = threadIdx.x; t < 4096u; t += blockDim.x) dst[t] = 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 1024u; k += blockDim.x) dst[lut_hash[k]] = src[lut_slot[k]];
}
