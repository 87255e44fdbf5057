// snk_fast.hip.h -- 2-bit kernel for pure upper-case ACGT pairs (snk_fast_kernel, snk_fast_singles_kernel).
// Part of the device code of libsnacc_hip.so; see snk_common.hip.h for the execution model.
#pragma once
#include "snk_common.hip.h"

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
        SNK_TRACE_REC(1u, L.op, L.anchor, payload | (L.endcode << 28), L.iend);
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
    SNK_TRACE_REC(2u, cur, cand, (f << 24) | (e2 & 0xFFFFFFu), cur);
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

    // Block end, bail-out, or not started yet.  This return must come BEFORE the slide below: a
    // slide consumes w.nx, and only the refill further down restores the "nx = [rb+32, rb+48)"
    // invariant the next slide relies on.
    if (__builtin_expect(next > L.mfl1, 0))
        return snk_fast_block_step(L, T, tbl, bm, slot, out, status);

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
    if (__builtin_expect((o > 15u) | (!YONLY && cur > w.lim), 0)) {
        // long jump / source change / seam: re-seat the reservoir
        if (cur >= L.s.lx + 4u)                          snk_win_init(w, L.s.arena, L.s.yoff, L.s.lx, 0xFFFFFFFFu, cur);
        else if (cur >= 4u && cur + 12u <= L.s.lx)       snk_win_init(w, L.s.arena, L.s.xoff, 0u, L.s.lx - 12u, cur);
        else                                             { w.lim = 0u; wslow = true; }
        o = cur - 4u - w.rb;
    }
    SNK_TRACE_REC(8u, w.r0, w.r1, w.nx, cur);
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
    SNK_TRACE_REC(4u, cur, cand, (f << 24) | (m ? 0x800000u : 0u) | (valid ? 0x400000u : 0u) | (e2 & 0x3FFFFFu), cur);
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
#ifdef SNK_HOST_EMU
    const uint16_t *const lut0 = slot;
#else
    const SNK_AS3 uint16_t *const lut0 = (const SNK_AS3 uint16_t *)0;
#endif

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
            SNK_TRACE_REC(3u, cur, cand, (f << 24) | (m ? 0x800000u : 0u) | (valid ? 0x400000u : 0u) | (e2 & 0x3FFFFFu), cur);
            SNK_TRACE_REC(5u, wc, w.rb, nxoff - w.soff, cur);
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
#ifndef SNK_HOST_EMU
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
#endif
    uint16_t *slot = (uint16_t *)snk_lds8;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t chains = lanes * waves;

    for (uint32_t t = tid; t < 512u; t += SNK_COOP(blockDim.x))
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
        for (uint32_t t = lane; t < SNK_FSLOTS / 2u; t += SNK_COOP(64u)) {
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
        for (uint32_t t = lane; t < SNK_FBMWORDS; t += SNK_COOP(64u)) ((uint32_t *)(dst + SNK_FSLOTS * 2u))[t] = use ? 0u : 0xFFFFFFFFu;
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
