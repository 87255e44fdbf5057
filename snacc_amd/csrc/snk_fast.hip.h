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
    snk_g8 *marena;           // ... of the class arena (sequences with exceptions; same layout)
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

// The same window from the class arena (2-bit class per base: 00 the set's letters, 01 the other case, 11 another byte).
__device__ __forceinline__ uint32_t snk_fetch32m(const SnkFastSrc &s, uint32_t p)
{
    SnkFastSrc m = s;
    m.arena = s.marena;
    return snk_fetch32(m, p);
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
    uint32_t mfl1, mlimit, olimit;
    uint32_t base;                         // virtual base of the block: its start minus k3; the table stores cur - base
    uint32_t k3;                           // -lx mod 4: makes (base - lx - 4) a multiple of 4 (see snk_fast_steady)
    uint32_t endcode;                      // 0 running, 1 ends with last-literals, 2 liblz4 gave up (raw)
    bool pending;                          // put(cur-2) owed before the next probe
    SnkWin w;
    // ---- only used by the instantiations for sequences with exceptions (EXC) ----
    SnkGenSrc g;                           // the ASCII bytes of x and y
    const uint32_t *fx, *fy;               // dilated exception flags of x / y (NULL: none): any position, one load, conservative
    const uint32_t *rx, *ry;               // exact exception runs of x / y ({start, end} pairs, NULL: none): around the cursor
    uint32_t ri;                           // first run of the cursor's sequence that does not end before the cursor window
    bool ron_y;                            // ... which sequence that is
    uint32_t *ovf;                         // this chain's overflow table
    uint32_t xlim;                         // first position >= the last scan whose window is not clean
    uint32_t site_lo, site_hi;             // stream positions p in [site_lo, site_hi) are known to be unclean (the run the last
                                           // scan found the cursor at): no look at the run list for them
    uint32_t mask_until;                   // cursors below this may read a table entry whose window holds an exception: the
                                           // steady loop then runs with the mask window (see "Exceptions" below)
    uint32_t olo, olim;                    // cursors in [olo, olim) have a window of other-case letters only (class 01): the
                                           // steady loop's other-case mode serves them (see snk_exc_other_ready); olo = olim: unknown
    uint32_t oscan;                        // cursor of the last scan that found no such range (not scanned again there)
};

// diagnostic build only (-DSNK_STATS, `make stats`): event counters of the exception machinery (tools/gpu_exc.py) and a
// cycle account of the waves (tools/gpu_account.py): 0..6 events, 7 wave cycles, 13 cycles inside the steady loop, 14 loop
// entries, 15 trips of the C++ loop (same trips as the asm loop's), 16..22 why lanes asked for service, 24 finish,
// 25 / 26 general-probe rounds between two loop entries (cycles / rounds), 27 loop prologue.
#ifdef SNK_STATS
__device__ unsigned long long snk_stats[64];      // [32..63]: the same account for the far waves
#define SNK_COUNT(i) atomicAdd(&snk_stats[i], 1ull)
// per-lane events of the general path (one atomic per probe and lane: they slow a run with many exception sites down
// several times over; -DSNK_STATS=2 counts them, the plain stats build keeps the cycle account undistorted)
#if SNK_STATS + 0 >= 2
#define SNK_COUNT_HEAVY(i) SNK_COUNT(i)
#else
#define SNK_COUNT_HEAVY(i) do { } while (0)
#endif
// the cycle account is kept in registers and added to snk_stats once, when the wave ends (atomics on the way would
// change what they measure): every lane carries the same numbers, lane 0 reports them
struct SnkProf { unsigned long long loop, finish, rounds_cyc, probe, prologue, top, other, otrips, olanes, oin, oout, be_cyc; unsigned int entries, rounds, jobs, oruns, oswaps, be_n, be_rounds; };
#define SNK_PROF_ARG , SnkProf &P
#define SNK_PROF_PASS , P
#else
#define SNK_PROF_ARG
#define SNK_PROF_PASS
#define SNK_COUNT(i) do { } while (0)
#define SNK_COUNT_HEAVY(i) do { } while (0)
#endif

// =========================================================================
//  Exceptions: a few non-ACGT bytes in an otherwise 2-bit sequence
// =========================================================================
// A position p is CLEAN when its window [p-4, p+12) lies inside one sequence, >= 4 bases from the
// stream start, and its 16-base granule is not flagged (no exception within the window, by the
// dilation of the flags).  The steady loop only ever probes and inserts at clean CURSOR positions:
// 5-mer and hash are then exactly what liblz4 sees on the bytes.  Every other cursor position goes
// through the byte-accurate general path below: hash of the real 5 bytes; if an ACGT 5-mer has that
// hash the slot of the 2-bit table takes the position (as an ordinary 16-bit offset), otherwise
// ovf[hash] does (the chain's overflow table: liblz4's own table restricted to the non-ACGT hashes).
// CANDIDATES may be anywhere: beside the packed arena lies a mask arena of the same layout (2 bits per
// base, 11 where the byte is not one of ACGT), and the steady loop ORs the candidate's mask window
// into the difference of the 2-bit windows: an exception in the candidate window ends the match
// there, exactly as the byte compare does (the cursor window holds ACGT only, an exception byte
// equals none of them).  Round 2's first version wrote a sentinel instead of the offset when the
// inserted position was not clean; the ~3.5 service exits per lane and site that the later reads
// of those sentinels cost were most of a site's price.
__device__ __forceinline__ bool snk_exc_clean(const SnkFastLane &L, uint32_t p)
{
    const uint32_t lx = L.s.lx;
    if (p + 12u <= lx) {                                        // (p < 4: the bases before the stream start are zero padding,
        if (!L.fx) return true;                                 //  and the steady loop never extends a match back beyond position 0)
        const uint32_t g = p < 4u ? 0u : (p - 4u) >> 4;
        return ((L.fx[g >> 5] >> (g & 31u)) & 1u) == 0u;
    }
    if (p >= lx + 4u) {
        if (!L.fy) return true;
        const uint32_t g = (p - 4u - lx) >> 4;
        return ((L.fy[g >> 5] >> (g & 31u)) & 1u) == 0u;
    }
    return false;                                               // the window straddles the seam
}

// Exact versions for positions around the cursor, from the sorted run list of the cursor's sequence: the
// window [q-4, q+12) of sequence position q touches the run [a, b) iff a < q + 12 and b > q - 4.  L.ri only
// moves forward (the cursor does), so both are O(1) amortised -- no scan of the flag words.
__device__ __forceinline__ void snk_exc_seek(SnkFastLane &L, uint32_t p)
{
    const bool ony = p >= L.s.lx;
    if (ony != L.ron_y) { L.ron_y = ony; L.ri = 0u; }
    const uint32_t *r = ony ? L.ry : L.rx;
    if (!r) return;
    const uint32_t q = p - (ony ? L.s.lx : 0u);
    while (r[2u * L.ri + 1u] != 0xFFFFFFFFu && r[2u * L.ri + 1u] + 4u <= q) L.ri++;          // runs that end before [q-4, ...)
}

// is the window of position `pos` (the cursor, or a few bases behind it) clean?  Call snk_exc_seek(L, cursor) first.
__device__ __forceinline__ bool snk_exc_clean_near(const SnkFastLane &L, uint32_t pos)
{
    const uint32_t lx = L.s.lx;
    const bool inx = pos + 12u <= lx, iny = pos >= lx + 4u;
    if (!(inx | iny)) return false;                              // the window straddles the seam
    const uint32_t *r = iny ? L.ry : L.rx;
    if (!r || iny != L.ron_y) return r == nullptr;               // (a position in the other sequence than the cursor's: say no)
    const uint32_t q = pos - (iny ? lx : 0u);
    for (uint32_t k = L.ri > 0u ? L.ri - 1u : 0u; r[2u * k] != 0xFFFFFFFFu && r[2u * k] < q + 12u; ++k)
        if (r[2u * k + 1u] + 4u > q) return false;
    return true;
}

// first position p' >= p whose window is not clean (0xFFFFFFFF: none); p is the cursor
__device__ __forceinline__ uint32_t snk_exc_next(SnkFastLane &L, uint32_t p)
{
    const uint32_t lx = L.s.lx;
    if (p - L.site_lo < L.site_hi - L.site_lo) return p;         // still at the run found last time
    snk_exc_seek(L, p);
    if (p + 12u <= lx) {
        const uint32_t stop = lx - 11u;                         // the seam gap at the latest
        if (!L.rx) return stop;
        const uint32_t a = L.rx[2u * L.ri];                      // first run that does not end before the window
        if (a == 0xFFFFFFFFu) return stop;
        if (a < p + 12u) { L.site_lo = a > 11u ? a - 11u : 0u; L.site_hi = L.rx[2u * L.ri + 1u] + 4u; }
        const uint32_t q = a < p + 12u ? p : a - 11u;            // touching now, or from a - 11 on
        return q < stop ? q : stop;
    }
    if (p < lx + 4u) return p;
    if (!L.ry) return 0xFFFFFFFFu;
    const uint32_t a = L.ry[2u * L.ri];
    if (a == 0xFFFFFFFFu) return 0xFFFFFFFFu;
    if (a < (p - lx) + 12u) { L.site_lo = lx + (a > 11u ? a - 11u : 0u); L.site_hi = lx + L.ry[2u * L.ri + 1u] + 4u; }
    return a < (p - lx) + 12u ? p : lx + a - 11u;
}

__device__ __forceinline__ void snk_exc_put(SnkFastLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm, uint32_t h, uint32_t pos)
{
    const uint32_t s = T.lut_h2s[h];                                // < 0x8000: slot of the 2-bit table; else 0x8000 | index in the overflow table
    if (s < 0x8000u) {
        tbl[s] = (uint16_t)(pos - L.base);                          // (<= 65 527: puts end 12 bytes before the block does)
        atomicOr(&bm[s >> 5], 1u << (s & 31u));
    } else {
        SNK_COUNT_HEAVY(7);
        L.ovf[s & 0xFFFu] = pos;
    }
}

__device__ __forceinline__ uint32_t snk_exc_get(const SnkFastLane &L, const SnkTables &T, const uint16_t *tbl, const uint32_t *bm,
                                                uint32_t h, uint32_t cur, bool &valid)
{
    const uint32_t s = T.lut_h2s[h];
    if (s < 0x8000u) {
        const uint32_t e = tbl[s];
        const bool iscur = ((bm[s >> 5] >> (s & 31u)) & 1u) != 0u;
        valid = iscur | (e > cur - L.base);
        return L.base + e - (iscur ? 0u : 65536u);
    }
    const uint32_t cand = L.ovf[s & 0xFFFu];
    valid = cand + SNK_MAXDIST >= cur;
    return cand;
}

// liblz4's match accounting on the real bytes (cf. snk_fast_match_slow, which does it on 2-bit windows)
__device__ __forceinline__ void snk_exc_match(SnkFastLane &L, uint32_t cur, uint32_t cand)
{
    SNK_COUNT_HEAVY(4);
    const SnkGenSrc &g = L.g;
    uint32_t ip = cur;
    while (ip > L.anchor && cand > 0u && snk_byte_at(g, ip - 1u) == snk_byte_at(g, cand - 1u)) { ip--; cand--; }
    const uint32_t lit = ip - L.anchor;
    uint32_t op = L.op + 1u;
    bool bail = op + lit + 8u + lit / 255u > L.olimit;
    uint32_t a = ip + 4u;
    if (!bail) {
        op += lit + snk_lit_ext(lit) + 2u;
        uint32_t b = cand + 4u;
        while (a < L.mlimit) {
            const uint64_t d = snk_ld8(g, a) ^ snk_ld8(g, b);
            if (d) { a += (uint32_t)__builtin_ctzll(d) >> 3; break; }
            a += 8u; b += 8u;
        }
        if (a > L.mlimit) a = L.mlimit;
        const uint32_t mc = a - (ip + 4u);
        bail = op + 6u + (mc + 240u) / 255u > L.olimit;
        if (mc >= 15u) op += (mc - 15u) / 255u + 1u;
    }
    if (bail) { L.endcode = 2u; L.mfl1 = 0u; L.step = 1u; L.cur = cur; return; }
    L.op = op; L.anchor = a;
    L.cur = a; L.step = 1u; L.nb = 63u; L.pending = true;
    if (a >= L.mfl1) { L.endcode = 1u; L.mfl1 = 0u; }
}

// the byte-accurate counterpart of snk_fast_finish
__device__ __forceinline__ void snk_exc_finish(SnkFastLane &L, uint32_t cur, uint32_t cand, bool valid, uint64_t wc)
{
    const uint64_t wd = snk_ld8(L.g, valid ? cand : cur);
    if (valid & ((uint32_t)wc == (uint32_t)wd)) {
        snk_exc_match(L, cur, cand);
    } else {
        const uint32_t s3 = L.nb >> 6;
        L.cur = cur + L.step; L.step = s3 ? s3 : 1u; L.nb += 1u; L.pending = false;
    }
}

#if defined(SNK_STATS) && SNK_STATS + 0 >= 3
// diagnostic build only (STATS=3): where the general probe of a sequence with exceptions spends its cycles
// (snk_stats[40..47], by the first active lane; tools/gpu_exc.py prints them)
#define SNK_PSTAMP0 unsigned long long stp_ = clock64(); const bool stp_on_ = (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))
#define SNK_PSTAMP(i) do { const unsigned long long n_ = clock64(); if (stp_on_) atomicAdd(&snk_stats[40 + (i)], n_ - stp_); stp_ = n_; } while (0)
#else
#define SNK_PSTAMP0 do { } while (0)
#define SNK_PSTAMP(i) do { } while (0)
#endif

// ---- the general probe on (code, class) windows ---------------------------------------------------------------
// A byte of a 2-bit sequence is named exactly by its 2-bit code and its class unless the class is 11 (another byte: N,
// IUPAC codes).  So the general probe of a sequence with exceptions -- every cursor position of a soft-masked stretch --
// can hash, look up and compare on the packed and class arenas, which the wave's neighbours keep hot in the L1, instead
// of on the ASCII arena (cold lines, byte loops): the 5-mer's key comes from a LUT (the set's case: its slot; the other
// case: lut_okey), the candidate is compared as two 32-bit windows.  Whenever a class-11 byte could decide (inside the
// hashed 5 bytes, or where the match stops) the byte-accurate code above takes over, so the result is liblz4's in every case.
struct SnkKey { uint32_t s; uint32_t h; bool ok; };   // s < 896: slot of the 2-bit table; else ovf[h]; ok: the windows decided

__device__ __forceinline__ SnkKey snk_exc_key(const SnkTables &T, const uint16_t *slot, uint32_t code, uint32_t cls)
{
    SnkKey k; k.h = 0u; k.ok = true;
    if (cls == 0u)          k.s = slot[code];                          // five letters of the set's case
    else if (cls == 0x155u) {                                          // ... of the other case
        const uint32_t v = T.lut_okey[code];
        k.s = v < SNK_FSLOTS ? v : 0xFFFFu; k.h = v & 0xFFFu;
    } else { k.s = 0xFFFFu; k.ok = false; }                            // mixed, or another byte: hash the real bytes
    return k;
}

__device__ __forceinline__ void snk_exc_put_key(SnkFastLane &L, uint16_t *tbl, uint32_t *bm, const SnkKey k, uint32_t pos)
{
    if (k.s != 0xFFFFu) {
        tbl[k.s] = (uint16_t)(pos - L.base);
        atomicOr(&bm[k.s >> 5], 1u << (k.s & 31u));
    } else {
        L.ovf[k.h] = pos;
    }
}

__device__ __forceinline__ uint32_t snk_exc_get_key(const SnkFastLane &L, const uint16_t *tbl, const uint32_t *bm, const SnkKey k,
                                                    uint32_t cur, bool &valid)
{
    if (k.s != 0xFFFFu) {
        const uint32_t e = tbl[k.s];
        const bool iscur = ((bm[k.s >> 5] >> (k.s & 31u)) & 1u) != 0u;
        valid = iscur | (e > cur - L.base);
        return L.base + e - (iscur ? 0u : 65536u);
    }
    const uint32_t cand = L.ovf[k.h];
    valid = cand + SNK_MAXDIST >= cur;
    return cand;
}

// positions (both bits of the base's pair) whose class is 11
__device__ __forceinline__ uint32_t snk_c11(uint32_t m)
{
    const uint32_t b = m & (m >> 1) & 0x55555555u;
    return b | (b << 1);
}

// Finish the probe at `cur` (table operations done, candidate `cand`) on the windows wc / wk (codes / classes at cur).
// Returns false when a class-11 byte could decide: the caller then finishes on the real bytes.
__device__ __forceinline__ bool snk_exc_finish_win(SnkFastLane &L, uint32_t cur, uint32_t cand, bool valid, uint32_t wc, uint32_t wk)
{
    if (!valid) {
        const uint32_t s3 = L.nb >> 6;
        L.cur = cur + L.step; L.step = s3 ? s3 : 1u; L.nb += 1u; L.pending = false;
        return true;
    }
    const uint32_t wd = snk_fetch32(L.s, cand), wdk = snk_fetch32m(L.s, cand);
    // diff: the bytes differ for certain (codes differ, or classes differ -- a class-11 byte is never a letter);
    // amb: both bytes are of class 11 with equal codes (N and N, N and R, ...): only the real bytes can tell
    const uint32_t diff = (wc ^ wd) | (wk ^ wdk), amb = snk_c11(wk) & snk_c11(wdk);
    const uint32_t x = diff | amb;
    const uint32_t f = (uint32_t)__builtin_ctz((x >> 8) | (1u << 24)) >> 1;              // bases from cur up to the first event, 0..12
    if (f < 12u && ((diff >> (8u + 2u * f)) & 3u) == 0u) return false;                    // the event is a pair of class-11 bytes: bytes decide
    if (f < 4u) {                                                                         // the first four bytes differ: no match
        const uint32_t s3 = L.nb >> 6;
        L.cur = cur + L.step; L.step = s3 ? s3 : 1u; L.nb += 1u; L.pending = false;
        return true;
    }
    const uint32_t eq = (uint32_t)__builtin_clz(((x & 0xFFu) << 24) | 0x00800000u) >> 1;  // equal bases before cur, 0..4
    uint32_t lit = cur - L.anchor;
    uint32_t b = eq < lit ? eq : lit;
    b = b < cand ? b : cand;
    if (b == eq && eq < 4u && ((diff >> (6u - 2u * eq)) & 3u) == 0u) return false;        // the back-extension stops at a pair of class-11 bytes
    lit -= b;
    const uint32_t e2 = cur + f, opn = L.op + lit + 3u;
    if ((f < 12u) & (b < 4u) & (lit < 15u) & (opn + 6u <= L.olimit) & (e2 < L.mfl1)) {
        L.op = opn; L.anchor = e2; L.cur = e2; L.step = 1u; L.nb = 63u; L.pending = true;
    } else {
        snk_exc_match(L, cur, cand);                                                      // may run on: liblz4's loops on the real bytes
    }
    return true;
}

// ---- other-case stretches (soft-masked genomes) ------------------------------------------------------------------
// Inside a stretch of the other case every cursor window is "unclean" for the steady loop proper, but it is as regular
// as the rest of the genome: letters only, all of one case.  Such cursors run a second mode of the steady loop (OTH, C++
// statement only): the 5-mer's key comes from lut_okey (liblz4's hash of the other-case 5-mer: a slot of the 2-bit
// table where a 5-mer of the set's case shares the hash, else the chain's overflow table), the candidate's class
// window must be 01 throughout (XOR with 0x55555555 joins the difference), everything else is the loop as it is.
// [olo, olim): cursors whose window [p-4, p+12) lies inside one sequence and holds class 01 only, found by a scan of the
// class arena from the cursor on (up to 4096 bases at a time).
__device__ __forceinline__ bool snk_exc_other_ready(SnkFastLane &L)
{
    const uint32_t cur = L.cur, lx = L.s.lx;
    if (cur - L.olo < L.olim - L.olo) return true;
    if (cur == L.oscan) return false;
    L.oscan = cur; L.olo = L.olim = 0u;
    const bool iny = cur >= lx + 4u;
    if (!iny && !(cur >= 4u && cur + 12u <= lx)) return false;             // seam gap, stream start
    if (snk_fetch32m(L.s, cur) != 0x55555555u) return false;
    const uint32_t org = iny ? lx : 0u, soff = iny ? L.s.yoff : L.s.xoff;
    uint32_t q = cur - org + 12u;                                           // first base behind the cursor's window
    const uint32_t qend = q + 4096u;
    uint32_t bad = 0xFFFFFFFFu;
    while (q < qend) {
        const uint64_t v = snk_ld8g(L.s.marena + (size_t)(soff + (q >> 2))) ^ 0x5555555555555555ull;      // 32 bases from q & ~3
        const uint64_t m = v >> (2u * (q & 3u));                            // base q at bit 0 (behind the end: class 00 -> set)
        if (m) { bad = q + ((uint32_t)__builtin_ctzll(m) >> 1); break; }
        q += 32u - (q & 3u);
    }
    L.olo = cur;
    L.olim = org + (bad != 0xFFFFFFFFu ? bad : q) - 11u;                    // first cursor whose window reaches the bad base
    return L.olim > cur;
}


// ---- the other-case table of a chain in its LDS region, for the length of a stretch (round 4) ------------------------------
// Inside a stretch of the other case every probe keys on an other-case 5-mer: 1024 codes, 895 (894) distinct liblz4 hashes,
// numbered 0 .. 894 by T.lut_oj.  About a quarter of those hashes are also hashes of set-case 5-mers (their entries live in the
// chain's 2-bit table), the others live in the chain's overflow table in global memory -- where round 3's other-case mode read
// and wrote them once per probe: four dependent global accesses per trip, 159 GB of HBM traffic per launch at 5 % lower case
// (profiles/r04_pmc_traffic_softmask5.json).  Here the lanes gathered at a stretch swap tables instead: the set-case table
// goes out to the chain's save area (raw copy), the other-case entries come in -- shared ones from the table itself, the
// others from the overflow table, converted to the table's (16-bit offset, written-this-block bit) form, which states liblz4's
// distance rule exactly as for every other entry -- the stretch is walked by the hand-scheduled loop on a table of the usual
// layout with the other case's LUT, and at the end everything goes back where the general path expects it.  No block edge is
// crossed inside the mode (the loop's limit includes the block's), so the conversions use one base.
// All lanes of the wave work on one chain at a time (`lane`: 0 .. 63; the CPU emulation's one lane walks the loops alone).
#define SNK_OTH_SLOTS 895u
// (What the swaps cost decides whether the mode pays: the first form -- one chain at a time, the map and the overflow entries
// loaded inside the per-chain loops -- took 312 k + 264 k cycles per stretch and wave, six dependent global latencies per chain.
// Here the map is read once per swap, and the overflow-table loads / save-area loads of the NEXT chain are in flight while the
// current chain is converted.)
struct SnkOthChain { uint8_t *region; uint32_t *ov; uint32_t *sv; uint32_t base; };
#define SNK_OTH_STR  SNK_COOP(64u)
#define SNK_OTH_NPL  ((SNK_FSLOTS + SNK_OTH_STR - 1u) / SNK_OTH_STR)            /* entries per lane: 14 (the emulation's one lane: 896) */
#define SNK_OTH_NSV  ((SNK_FCHAIN_B / 4u + SNK_OTH_STR - 1u) / SNK_OTH_STR)     /* words of a table per lane: 8 */

// the chain of lane l of the wave (wave-uniform description, every lane gets the same)
template <bool SPEC>
__device__ __forceinline__ SnkOthChain snk_oth_chain(const SnkTables &T, uint8_t *lds, uint32_t l, uint32_t flut, uint32_t wave, uint32_t lanes,
                                                     size_t chain0, uint32_t base_of_lane)
{
    const uint32_t lc = SPEC ? l >> 1 : l;
    SnkOthChain c;
    c.region = lds + flut + (size_t)(wave * lanes + lc) * SNK_FCHAIN_B;
    c.ov = T.ovf + (chain0 + lc) * 4096u;
    c.sv = T.osave + (chain0 + lc) * 512u;
    c.base = (uint32_t)__shfl((int)base_of_lane, (int)l);
    return c;
}

template <bool SPEC>
__device__ __forceinline__ void snk_oth_swap_in_all(const SnkTables &T, uint8_t *lds, unsigned long long tm, uint32_t flut, uint32_t wave, uint32_t lanes,
                                                    size_t chain0, uint32_t my_base, uint32_t lane)
{
    uint32_t om[SNK_OTH_NPL];                                  // this lane's entries of the map: hash | shared slot << 16
#pragma unroll
    for (uint32_t i = 0; i < SNK_OTH_NPL; ++i) {
        const uint32_t j = lane + SNK_OTH_STR * i;
        om[i] = j < SNK_OTH_SLOTS ? T.lut_omap[j] : 0u;
    }
    // (1) every set-case table out (raw copies; nothing waits for them)
    for (unsigned long long t2 = tm; t2; t2 &= t2 - 1ull) {
        const SnkOthChain c = snk_oth_chain<SPEC>(T, lds, (uint32_t)__builtin_ctzll(t2), flut, wave, lanes, chain0, my_base);
        const uint32_t *r32 = (const uint32_t *)c.region;
        for (uint32_t t = lane; t < SNK_FCHAIN_B / 4u; t += SNK_OTH_STR) c.sv[t] = r32[t];
    }
    // (2) the other-case entries in: shared slots from the table itself, the others from the overflow table.  pc[i] holds the
    // overflow entry of THIS chain until it is converted and is then given the load of the NEXT chain's, which is in flight
    // while this chain's table is written (one register set: the kernel must not spill)
    uint32_t pc[SNK_OTH_NPL];
    {
        const SnkOthChain c = snk_oth_chain<SPEC>(T, lds, (uint32_t)__builtin_ctzll(tm), flut, wave, lanes, chain0, my_base);
#pragma unroll
        for (uint32_t i = 0; i < SNK_OTH_NPL; ++i) pc[i] = c.ov[lane + SNK_OTH_STR * i];          // (entry j of the overflow table: coalesced)
    }
    for (unsigned long long t2 = tm; t2; t2 &= t2 - 1ull) {
        const SnkOthChain c = snk_oth_chain<SPEC>(T, lds, (uint32_t)__builtin_ctzll(t2), flut, wave, lanes, chain0, my_base);
        const unsigned long long t3 = t2 & (t2 - 1ull);
        const SnkOthChain n = snk_oth_chain<SPEC>(T, lds, (uint32_t)__builtin_ctzll(t3 ? t3 : t2), flut, wave, lanes, chain0, my_base);
        uint16_t *const tb = (uint16_t *)c.region;
        uint32_t *const bmw = (uint32_t *)(c.region + SNK_FSLOTS * 2u);
        uint32_t val[SNK_OTH_NPL];                             // offset | bit << 16
        // (branch-free: both sources are read for every entry -- a shared slot's index is a valid index of the overflow table and the
        // other way round --, the map decides by a select; the loads of the next chain go out as soon as this chain's are used)
#pragma unroll
        for (uint32_t i = 0; i < SNK_OTH_NPL; ++i) {
            const uint32_t j = lane + SNK_OTH_STR * i, sl = om[i] >> 16, sx = sl < SNK_FSLOTS ? sl : 0u;
            const uint32_t vs = tb[sx] | (((bmw[sx >> 5] >> (sx & 31u)) & 1u) << 16);
            const int32_t rel = (int32_t)(pc[i] - c.base);                             // absolute position -> the table's form
            const uint32_t vo = rel >= 0 ? ((uint32_t)rel | 0x10000u) : rel >= -65536 ? (uint32_t)(rel + 65536) : 0u;
            pc[i] = n.ov[j];                                                           // the next chain's entry (the last chain: its own again)
            val[i] = j < SNK_OTH_SLOTS ? (sl != 0xFFFFu ? vs : vo) : 0u;
        }
        for (uint32_t t = lane; t < SNK_FBMWORDS; t += SNK_OTH_STR) bmw[t] = 0u;      // (after every read of the old table)
#pragma unroll
        for (uint32_t i = 0; i < SNK_OTH_NPL; ++i) {
            const uint32_t j = lane + SNK_OTH_STR * i;
            if (j < SNK_FSLOTS) {
                tb[j] = (uint16_t)val[i];
                if (val[i] >> 16) atomicOr(&bmw[j >> 5], 1u << (j & 31u));
            }
        }
    }
}

template <bool SPEC>
__device__ __forceinline__ void snk_oth_swap_out_all(const SnkTables &T, uint8_t *lds, unsigned long long tm, uint32_t flut, uint32_t wave, uint32_t lanes,
                                                     size_t chain0, uint32_t my_base, uint32_t lane)
{
    uint32_t om[SNK_OTH_NPL];
#pragma unroll
    for (uint32_t i = 0; i < SNK_OTH_NPL; ++i) {
        const uint32_t j = lane + SNK_OTH_STR * i;
        om[i] = j < SNK_OTH_SLOTS ? T.lut_omap[j] : 0u;
    }
    uint32_t sc[SNK_OTH_NSV];                                  // the saved set-case table of the current chain, then the next one's loads
    {
        const SnkOthChain c = snk_oth_chain<SPEC>(T, lds, (uint32_t)__builtin_ctzll(tm), flut, wave, lanes, chain0, my_base);
#pragma unroll
        for (uint32_t k = 0; k < SNK_OTH_NSV; ++k) { const uint32_t t = lane + SNK_OTH_STR * k; sc[k] = t < SNK_FCHAIN_B / 4u ? c.sv[t] : 0u; }
    }
    for (unsigned long long t2 = tm; t2; t2 &= t2 - 1ull) {
        const SnkOthChain c = snk_oth_chain<SPEC>(T, lds, (uint32_t)__builtin_ctzll(t2), flut, wave, lanes, chain0, my_base);
        const unsigned long long t3 = t2 & (t2 - 1ull);
        const SnkOthChain n = snk_oth_chain<SPEC>(T, lds, (uint32_t)__builtin_ctzll(t3 ? t3 : t2), flut, wave, lanes, chain0, my_base);
        uint32_t *const r32 = (uint32_t *)c.region;
        uint16_t *const tb = (uint16_t *)c.region;
        uint32_t *const bmw = (uint32_t *)(c.region + SNK_FSLOTS * 2u);
        uint32_t val[SNK_OTH_NPL];
#pragma unroll
        for (uint32_t i = 0; i < SNK_OTH_NPL; ++i) {                                     // the other-case entries: to the overflow table ...
            const uint32_t j = lane + SNK_OTH_STR * i;
            uint32_t v = 0u;
            if (j < SNK_OTH_SLOTS) {
                const uint32_t e = tb[j], b = (bmw[j >> 5] >> (j & 31u)) & 1u;
                v = e | (b << 16);
                // (an entry that is neither of this block nor in reach from it is dead for good: positions only grow.  Dead entries are
                // zeroed at block ends from the third block on, where position 0 is out of reach too -- so 0 says the same)
                if ((om[i] >> 16) == 0xFFFFu) c.ov[j] = b ? c.base + e : (e ? c.base - 65536u + e : 0u);
            }
            val[i] = v;
        }
#pragma unroll
        for (uint32_t k = 0; k < SNK_OTH_NSV; ++k) {                                     // the set-case table back (and the next chain's on its way)
            const uint32_t t = lane + SNK_OTH_STR * k;
            if (t < SNK_FCHAIN_B / 4u) { r32[t] = sc[k]; sc[k] = n.sv[t]; }
        }
#pragma unroll
        for (uint32_t i = 0; i < SNK_OTH_NPL; ++i) {                                     // ... or, shared hashes, into its slots (a slot put inside
            const uint32_t j = lane + SNK_OTH_STR * i, sl = om[i] >> 16;                 // the stretch carries its bit; an untouched one is what it was)
            if (j < SNK_OTH_SLOTS && sl != 0xFFFFu && (val[i] >> 16)) { tb[sl] = (uint16_t)val[i]; atomicOr(&bmw[sl >> 5], 1u << (sl & 31u)); }
        }
    }
}

struct alignas(16) SnkWord4 { uint32_t a, b, c, d; };
// Rare path (once per 64 KiB): close the finished block, age the table, open the next block.
// Returns true when the frame is complete (size written).
// FAR (chains of the extra waves, see "Chains beyond the LDS" below): the table is gt[896], absolute positions as u32 in
// global memory -- no bitmap, no aging (a candidate is in range when cur - e <= 65535); tbl / bm are unused then.
template <bool EXC, bool FAR>
__device__ __forceinline__ bool snk_fast_block_step(SnkFastLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm, uint32_t *gt,
                                                 const uint16_t *slot, uint32_t *out, uint32_t *status)
{
#if SNK_STATS + 0 >= 3
    unsigned long long bs_ = clock64();
#define SNK_BSTAMP(i) do { const unsigned long long n_ = clock64(); atomicAdd(&snk_stats[i], n_ - bs_); bs_ = n_; } while (0)
#else
#define SNK_BSTAMP(i) do { } while (0)
#endif
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
        if (!FAR && L.snap != 0 && L.pos == L.spos && L.spos != 0u) {
            // prefix snapshot: absolute positions; entries older than one block -> 0 (too far for good)
            uint32_t *dst = T.snap_fast + (size_t)L.xi * SNK_FSLOTS;
            for (uint32_t t = 0; t < SNK_FSLOTS; ++t) {
                uint32_t v = ((bm[t >> 5] >> (t & 31u)) & 1u) ? (L.base + tbl[t]) : 0u;
                dst[t] = v;
            }
            if (EXC) {      // liblz4's whole table (what a byte kernel, or another chain with exceptions, starts from)
                uint32_t *gen = T.snap_gen + (size_t)L.xi * 4096u;
                for (uint32_t h = 0; h < 4096u; ++h) {
                    const uint32_t sl = T.lut_h2s[h];
                    gen[h] = sl < 0x8000u ? dst[sl] : L.ovf[sl & 0xFFFu];
                }
            }
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
        SNK_BSTAMP(37);
        // (pure ACGT chains) the reservoir of the new block's first cursor is seated HERE, its three loads in flight while the
        // table is aged -- when that cursor reads the source the reservoir already reads (15 of a pair's 16 block ends lie inside
        // y); the 5-mer of the block's first position then comes out of the reservoir too.  A block end cost the wave 7.4 k cycles
        // between two loop entries where any other exit costs 2.8 k: 28 dependent LDS reads of the ageing, one global load for
        // that 5-mer, and a further round for the re-seat (profiles/r04_cycle_account.json, block_end_passes).
        bool seated = false;
        if (!EXC && !FAR) {
            const uint32_t ncur = L.pos + 1u, lx = L.s.lx;
            if (ncur >= lx + 4u && L.w.org == lx && L.w.soff == L.s.yoff && L.w.lim == 0xFFFFFFFFu) {
                snk_win_init(L.w, L.s.arena, L.s.yoff, lx, 0xFFFFFFFFu, ncur); seated = true;
            } else if (ncur >= 4u && ncur + 12u <= lx && L.w.org == 0u && L.w.soff == L.s.xoff && L.w.lim == lx - 12u) {
                snk_win_init(L.w, L.s.arena, L.s.xoff, 0u, lx - 12u, ncur); seated = true;
            }
        }
        if (!FAR && !L.first) {
            // age the table: entries not written during the block just finished are dead.  (Four bitmap words per LDS read -- the
            // chain's region is 16-byte aligned -- and, where the registers are there, all seven reads issued before the first is
            // used: word by word this was 28 dependent LDS round trips.)
            static_assert(SNK_FBMWORDS % 4u == 0u && SNK_FCHAIN_B % 16u == 0u && (SNK_FSLOTS * 2u) % 16u == 0u, "bitmap rows of 16 bytes");
            constexpr uint32_t ROWS = SNK_FBMWORDS / 4u, RB = EXC ? 1u : ROWS;      // (the kernels for exceptions have no registers to spare)
            // (With one wave per SIMD every instruction of this path costs the wave ~5 cycles, so the common case is made short:
            // after a whole block nearly every slot has been written -- a row of 128 slots whose bits are all set costs three ANDs
            // and a compare.)
            for (uint32_t q0 = 0; q0 < ROWS; q0 += RB) {
                SnkWord4 v[RB];
#pragma unroll
                for (uint32_t r = 0; r < RB; ++r) v[r] = *(const SnkWord4 *)(bm + 4u * (q0 + r));
#pragma unroll
                for (uint32_t r = 0; r < RB; ++r) {
                    if ((v[r].a & v[r].b & v[r].c & v[r].d) == 0xFFFFFFFFu) continue;
                    const uint32_t w4[4] = { v[r].a, v[r].b, v[r].c, v[r].d };
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) {
                        uint32_t z = ~w4[k];
                        while (z) {
                            const uint32_t b = (uint32_t)__builtin_ctz(z);
                            tbl[(4u * (q0 + r) + k) * 32u + b] = 0;
                            z &= z - 1u;
                        }
                    }
                }
                const SnkWord4 zero = { 0u, 0u, 0u, 0u };
#pragma unroll
                for (uint32_t r = 0; r < RB; ++r) *(SnkWord4 *)(bm + 4u * (q0 + r)) = zero;
            }
        }
        SNK_BSTAMP(38);
        L.first = false;
        L.base = L.pos - L.k3;
        L.mfl1 = L.iend - 11u; L.mlimit = L.iend - 5u; L.olimit = L.blen - 1u;
        if (EXC) {
            snk_exc_seek(L, L.pos);
            if (!snk_exc_clean(L, L.pos)) L.mask_until = L.pos + 65536u;      // (an entry is read at most 65 535 bytes further on)
            snk_exc_put(L, T, tbl, bm, snk_hash5(snk_ld8(L.g, L.pos)), L.pos);
        } else if (FAR) {
            const uint32_t w0 = snk_fetch32(L.s, L.pos);
            gt[slot[(w0 >> 8) & 1023u]] = L.pos;
        } else {
            // the 5-mer at pos: bases pos - rb .. of the reservoir (rb = pos - 3 - ((pos - 3 - org) & 3): 3 .. 6 bases in)
            const uint32_t k5 = seated ? __builtin_amdgcn_alignbit(L.w.r1, L.w.r0, 2u * (L.pos - L.w.rb)) & 1023u
                                       : (snk_fetch32(L.s, L.pos) >> 8) & 1023u;
            const uint32_t s0 = slot[k5];
            tbl[s0] = (uint16_t)L.k3;                                 // the block start, relative to the virtual base
            atomicOr(&bm[s0 >> 5], 1u << (s0 & 31u));
        }
        L.cur = L.pos + 1u; L.step = 1u; L.nb = 64u; L.anchor = L.pos; L.op = 0u;
        L.pending = false; L.in_block = true;
        SNK_BSTAMP(39);
        return false;
    }
}

// Rare path of a match: long back-extension, long match, length-extension bytes, output
// budget, end of block -- liblz4's exact accounting.
#ifndef SNK_EXT_BASES
#define SNK_EXT_BASES 128u         // bases per step of a long match's extension (a multiple of 32)
#endif
// xb: the difference of the two windows' first four bases (cur-4 .. cur-1 against cand-4 .. cand-1, bits 0..7): the back-extension
// starts on them -- no load for the usual 0..3 bases; every exit of the loop has a lane or two in here, and the loop's first
// compare used to cost the wave a global round trip.
__device__ __forceinline__ void snk_fast_match_slow(SnkFastLane &L, uint32_t cur, uint32_t cand, uint32_t f,
                                                    uint32_t anchor0, uint32_t op0, uint32_t xb)
{
    const SnkFastSrc &s = L.s;
    uint32_t ip = cur, lit = cur - anchor0;
    {
        uint32_t k = (uint32_t)__builtin_clz((xb << 24) | 0x00800000u) >> 1;      // equal bases before the cursor inside the windows, 0..4
        k = k < lit ? k : lit;
        k = k < cand ? k : cand;
        ip -= k; cand -= k; lit -= k;
        if (k == 4u)
            while (ip > anchor0 && cand > 0u && snk_base_at(s, ip - 1u) == snk_base_at(s, cand - 1u)) { ip--; cand--; lit--; }
    }
    uint32_t e2 = cur + f;
    if (f == 12u) {
        // keep counting: SNK_EXT_BASES at a time where both runs lie inside one sequence, all loads of a step in flight together
        // (a tandem repeat is a match of hundreds of bases: at 16 bases per pair of dependent loads the Markov set spent 7.4 k
        // cycles per loop exit in here, profiles/r04_cycle_account_markov.json); 16 at a time across the seam.  (Reads beyond a
        // sequence stay inside the arena's slack; the count is clamped below.  Issuing the next step's loads ahead of this
        // step's compare was measured too: slower, on this set and on the LCG set -- the code grows in front of the loop.)
        uint32_t bpos = cand + (cur - ip) + 12u;
        const uint32_t lx = s.lx;
        while (e2 < L.mlimit) {
            const bool ey = e2 >= lx, by = bpos >= lx;
            uint32_t cnt, full;
            if ((ey || e2 + SNK_EXT_BASES <= lx) && (by || bpos + SNK_EXT_BASES <= lx)) {
                const uint32_t qe = e2 - (ey ? lx : 0u), qb = bpos - (by ? lx : 0u);
                snk_g8 *pe = s.arena + (size_t)((ey ? s.yoff : s.xoff) + (qe >> 2));
                snk_g8 *pb = s.arena + (size_t)((by ? s.yoff : s.xoff) + (qb >> 2));
                constexpr uint32_t NW = SNK_EXT_BASES / 32u;
                uint64_t ew[NW + 1u], bw[NW + 1u];
#pragma unroll
                for (uint32_t k = 0; k <= NW; ++k) { ew[k] = snk_ld8g(pe + 8u * k); bw[k] = snk_ld8g(pb + 8u * k); }
                const uint32_t she = 2u * (qe & 3u), shb = 2u * (qb & 3u);
                cnt = SNK_EXT_BASES;
#pragma unroll
                for (uint32_t k = NW; k-- > 0u; ) {                     // (from the far end: the nearest difference wins)
                    const uint64_t ea = she ? (ew[k] >> she) | (ew[k + 1u] << (64u - she)) : ew[k];
                    const uint64_t ba = shb ? (bw[k] >> shb) | (bw[k + 1u] << (64u - shb)) : bw[k];
                    const uint64_t d = ea ^ ba;
                    cnt = d ? 32u * k + ((uint32_t)__builtin_ctzll(d) >> 1) : cnt;
                }
                full = SNK_EXT_BASES;
            } else {
                const uint32_t d = snk_fetch32(s, e2 + 4u) ^ snk_fetch32(s, bpos + 4u);
                cnt = d ? ((uint32_t)__builtin_ctz(d) >> 1) : 16u;
                full = 16u;
            }
            e2 += cnt; bpos += cnt;
            if (cnt < full) break;
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

// Finish the probe at `cur` whose table operations are already done and whose candidate is `cand`
// (valid or not): general window fetches (any source, seam included), liblz4's exact match
// accounting.  This is where the steady loop hands over whenever a lane needs service.
// (a vote of the lanes that are active HERE; the emulation's lane is alone wherever it stands -- its pair exists only inside
// the two-lane loop)
#ifdef SNK_HOST_EMU
#define SNK_ALL_ACTIVE(p) (p)
#else
#define SNK_ALL_ACTIVE(p) __all(p)
#endif
__device__ __forceinline__ void snk_fast_finish(SnkFastLane &L, uint32_t cur, uint32_t cand, bool valid)
{
    // (both windows inside one sequence for every lane -- the rule away from the seam --: two loads in flight together; the
    // compiler makes a branch with its own wait of each snk_fetch32, one global round trip after the other at every exit)
    uint32_t wc, wd;
    {
        const uint32_t pd = valid ? cand : cur, lx = L.s.lx;
        const bool cy = cur >= lx + 4u, dy = pd >= lx + 4u;
        if (SNK_ALL_ACTIVE((cy || cur + 12u <= lx) && (dy || pd + 12u <= lx))) {
            const int32_t qc = (int32_t)cur - 4 - (cy ? (int32_t)lx : 0), qd = (int32_t)pd - 4 - (dy ? (int32_t)lx : 0);
            const uint64_t vc = snk_ld8g(L.s.arena + (size_t)(uint32_t)((int32_t)(cy ? L.s.yoff : L.s.xoff) + (qc >> 2)));
            const uint64_t vd = snk_ld8g(L.s.arena + (size_t)(uint32_t)((int32_t)(dy ? L.s.yoff : L.s.xoff) + (qd >> 2)));
            wc = __builtin_amdgcn_alignbit((uint32_t)(vc >> 32), (uint32_t)vc, (uint32_t)(qc & 3) * 2u);
            wd = __builtin_amdgcn_alignbit((uint32_t)(vd >> 32), (uint32_t)vd, (uint32_t)(qd & 3) * 2u);
        } else {
            wc = snk_fetch32(L.s, cur);
            wd = snk_fetch32(L.s, pd);
        }
    }
    const uint32_t x = wc ^ wd;
    const uint32_t f = (uint32_t)__builtin_ctz((x >> 8) | (1u << 24)) >> 1;          // equal bases from cur, 0..12
    SNK_TRACE_REC(6u, cur, cand, (f << 24) | (valid ? 0x400000u : 0u), cur);
    if (valid & (f >= 4u)) {
        snk_fast_match_slow(L, cur, cand, f, L.anchor, L.op, x & 0xFFu);
    } else {
        const uint32_t s3 = L.nb >> 6;
        L.cur = cur + L.step; L.step = s3 ? s3 : 1u; L.nb += 1u; L.pending = false;
    }
}

// One probe the general way (any source, any state): opens and closes blocks, walks the seam and
// the stream start, serves lanes the steady loop has handed over.  Returns true when the lane's
// frame is complete.  Rare path: block edges, seams, service -- speed is irrelevant here.
template <bool EXC, bool FAR>
__device__ __forceinline__ bool snk_fast_iter(SnkFastLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm, uint32_t *gt,
                                              const uint16_t *slot, uint32_t *out, uint32_t *status)
{
    const uint32_t cur = L.cur;
    if (cur + L.step > L.mfl1)                               // block end, bail-out, or not started yet
        return snk_fast_block_step<EXC, FAR>(L, T, tbl, bm, gt, slot, out, status);
    if (EXC) {                                               // liblz4's order on the real bytes: put(cur-2), get(cur), put(cur)
        SNK_COUNT_HEAVY(1);
        snk_exc_seek(L, cur);
        // the positions put below may have an exception in their windows (by the granule flags: a superset): the steady
        // loop then compares with the mask window for as long as such an entry can be read (65 535 bytes)
        // (a cursor at L.xlim stands at a site -- the caller keeps xlim at the first position >= the cursor whose window is
        // not clean -- and needs no look at the flags)
        if (cur >= L.xlim || !snk_exc_clean(L, cur) || (L.pending && !snk_exc_clean(L, cur - 2u))) { L.mask_until = cur + 65536u; SNK_COUNT_HEAVY(6); }
        // the windows of the cursor: codes and classes of the bases [cur-4, cur+12); 5-mer at cur = bits 8..17, at cur-2 = bits 4..13
        SNK_PSTAMP0;
        const uint32_t wc = snk_fetch32(L.s, cur), wk = snk_fetch32m(L.s, cur);
        SNK_PSTAMP(0);
        if (L.pending) {
            const SnkKey k2 = snk_exc_key(T, slot, (wc >> 4) & 1023u, (wk >> 4) & 1023u);
            if (k2.ok) snk_exc_put_key(L, tbl, bm, k2, cur - 2u);
            else       snk_exc_put(L, T, tbl, bm, snk_hash5(snk_ld8(L.g, cur - 2u)), cur - 2u);
        }
        SNK_PSTAMP(1);
        const SnkKey k1 = snk_exc_key(T, slot, (wc >> 8) & 1023u, (wk >> 8) & 1023u);
        if (k1.ok) {
            bool valid;
            SNK_PSTAMP(2);
            const uint32_t cand = snk_exc_get_key(L, tbl, bm, k1, cur, valid);
            snk_exc_put_key(L, tbl, bm, k1, cur);
            SNK_PSTAMP(3);
            if (!snk_exc_finish_win(L, cur, cand, valid, wc, wk))
                snk_exc_finish(L, cur, cand, valid, snk_ld8(L.g, cur));
            SNK_PSTAMP(4);
            return false;
        }
        const uint64_t wb = snk_ld8(L.g, cur);
        const uint32_t h = snk_hash5(wb);
        bool valid;
        const uint32_t cand = snk_exc_get(L, T, tbl, bm, h, cur, valid);
        snk_exc_put(L, T, tbl, bm, h, cur);
        snk_exc_finish(L, cur, cand, valid, wb);
        return false;
    }
    const uint32_t wc = snk_fetch32(L.s, cur);
    const uint32_t s1 = slot[(wc >> 8) & 1023u];             // slot of the 5-mer at cur
    uint32_t s2 = slot[(wc >> 4) & 1023u];                   // slot of the 5-mer at cur-2
    s2 = L.pending ? s2 : (SNK_FSLOTS - 1u);                 // nothing owed: aim the put at the unused slot
    if (FAR) {                                               // liblz4's own order on absolute positions
        gt[s2] = cur - 2u;
        const uint32_t e = gt[s1];
        gt[s1] = cur;
        snk_fast_finish(L, cur, e, cur - e <= SNK_MAXDIST);
        return false;
    }
    const uint32_t e = tbl[s1];
    const uint32_t bw = bm[s1 >> 5];
    const uint32_t c = cur - L.base;                         // offset from the virtual base
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
    snk_fast_finish(L, cur, cand, valid);
    return false;
}

// v_ffbl_b32: index of the lowest set bit, 0xFFFFFFFF for 0.
__device__ __forceinline__ uint32_t snk_ffbl(uint32_t v) { return v ? (uint32_t)__builtin_ctz(v) : 0xFFFFFFFFu; }

// (sequences with exceptions) the probe at cur -- table operations done, match not evaluated -- after the steady loop
__device__ __forceinline__ void snk_fast_exc_handover(SnkFastLane &L, const uint32_t cur, uint32_t cand, const bool valid)
{
    if (valid && !snk_exc_clean(L, cand)) {                   // an exception near the candidate: the real bytes decide
        SNK_COUNT_HEAVY(2);
        snk_exc_finish(L, cur, cand, valid, snk_ld8(L.g, cur));
    } else {
        // Cursor and candidate windows are clean here, so the 2-bit windows -- hot in the L1 -- decide the
        // ordinary cases exactly as the loop would; a match that may run on beyond them (back-extension 4,
        // 12 bases forward, budget, block end) is counted on the real bytes, which may hold an exception a
        // little further on.
        const uint32_t wc2 = snk_fetch32(L.s, cur);
        const uint32_t wd2 = snk_fetch32(L.s, valid ? cand : cur);
        const uint32_t x2 = wc2 ^ wd2;
        const uint32_t f = (uint32_t)__builtin_ctz((x2 >> 8) | (1u << 24)) >> 1;
        if (!(valid & (f >= 4u))) {
            const uint32_t s3 = L.nb >> 6;
            L.cur = cur + L.step; L.step = s3 ? s3 : 1u; L.nb += 1u; L.pending = false;
        } else {
            const uint32_t eq = (uint32_t)__builtin_clz(((x2 & 0xFFu) << 24) | 0x00800000u) >> 1;
            uint32_t lit = cur - L.anchor;
            uint32_t b = eq < lit ? eq : lit;
            b = b < cand ? b : cand;
            lit -= b;
            const uint32_t e2 = cur + f, opn = L.op + lit + 3u;
            if ((f < 12u) & (b < 4u) & (lit < 15u) & (opn + 6u <= L.olimit) & (e2 < L.mfl1)) {
                L.op = opn; L.anchor = e2; L.cur = e2; L.step = 1u; L.nb = 63u; L.pending = true;
            } else {
                snk_exc_match(L, cur, cand);
            }
        }
    }
}

// ---- the steady loop as gfx950 code (used by snk_fast_steady<true>) -----------------------------
// Temporaries live in v90..v126 (clobbered).  A lone wave issues one instruction every ~4.3 cycles,
// so a trip costs (instructions on the dependent chain) x 4.3 + the three latencies (table read,
// candidate load, LUT read) + whatever overflows their shadows.  Schedule: slot(cur) read first,
// owed put in its shadow; candidate load, and in ITS shadow the commit of the previous probe (op,
// anchor, reservoir) and everything that does not need the window; compare; next window + LUT
// reads; this probe's accounting and the one exit test in the LUT shadow (~14 instructions = the
// LUT latency).  gfx950 needs 2 wait states between a VALU that writes an SGPR/VCC and a VALU that
// reads it: every v_cmp has two instructions behind it before its consumer.
// Two instantiations: DUAL (candidate window from x or y, chosen per probe; windows straddling the
// seam trip the limit test) and YONLY (every lane's block lies wholly > 64 KiB past its seam).
// diagnostic builds only (-DSNK_PAD_x): 4 independent VALU instructions at one point of the loop;
// the trip time grows by ~17 cycles where issue is on the critical path and not at all inside an
// under-filled latency shadow (tools/gpu_padsweep.sh).
#ifndef SNK_PADN
#define SNK_PADN 4
#endif
#define SNK_STR2(x) #x
#define SNK_STR(x) SNK_STR2(x)
#define SNK_PAD4 ".rept " SNK_STR(SNK_PADN) "\n\tv_mov_b32_e32 v127, v127\n\t.endr\n\t"
#ifdef SNK_PAD_A
#define SNK_PADA SNK_PAD4
#else
#define SNK_PADA
#endif
#ifdef SNK_PAD_B
#define SNK_PADB SNK_PAD4
#else
#define SNK_PADB
#endif
#ifdef SNK_PAD_C
#define SNK_PADC SNK_PAD4
#else
#define SNK_PADC
#endif
#ifdef SNK_PAD_D
#define SNK_PADD SNK_PAD4
#else
#define SNK_PADD
#endif
#ifdef SNK_PAD_E
#define SNK_PADE SNK_PAD4
#else
#define SNK_PADE
#endif
// The asm contract, as a comment in the generated code (tests/test_asm_contract.py compiles this file with -S and reads it
// back): which registers the compiler gave the read-write operands and which the input-only ones.  The two sets must not
// meet -- the read-write operands are not early-clobber (that numbering runs 2 % slower: operand banks), so an input-only
// operand whose value the compiler can prove equal to a read-write operand's at entry may be given the SAME register, which
// the loop then overwrites (round 3 hit exactly that with a zero mask beside sl = 0).  No instruction, no cost.
#ifdef SNK_PARK
#define SNK_CONTRACT_BLK " %[blk]"
#define SNK_OPERAND_BLK , [blk] "s"(blk)
#else                                   /* (the parking mask exists in the diagnostic build only: no operand in the shipped loop) */
#define SNK_CONTRACT_BLK
#define SNK_OPERAND_BLK
#endif
// The head of every hand-scheduled loop sits on a 64-byte boundary (one instruction-cache line).  Left where it falls, the
// two-lane loop's rate moves by 2.3 % with whatever the code in front of it happens to measure: heads at 24 or 56 bytes past
// a 64-byte boundary are the slow ones, the six other multiples of 8 run alike.
// -DSNK_LOOP_P2ALIGN=0 builds the loops where they fall, -DSNK_LOOP_PADN=n puts n s_nop words behind the boundary (the sweep).
#ifndef SNK_LOOP_P2ALIGN
#define SNK_LOOP_P2ALIGN 6
#endif
#define SNK_LOOP_STR2(x) #x
#define SNK_LOOP_STR(x) SNK_LOOP_STR2(x)
#ifndef SNK_LOOP_PADN
#define SNK_LOOP_PADN 0
#endif
#define SNK_LOOP_ALIGN ".p2align " SNK_LOOP_STR(SNK_LOOP_P2ALIGN) "\n\t.fill " SNK_LOOP_STR(SNK_LOOP_PADN) ", 4, 0xbf800000\n\t"
#define SNK_STEADY_CONTRACT \
    "; snk-asm-contract inout %[c] %[wc] %[s1] %[s2] %[r0] %[r1] %[rbc] %[nxoff] %[anchor] %[op] %[opn] %[ns2] %[sm] %[sl]" \
    " | in %[lb] %[sx] %[kx] %[xoffB] %[yoffB] %[T0] %[limc] %[oz] %[dm] %[k8] %[arena] %[marena]" SNK_CONTRACT_BLK "\n\t"
#define SNK_STEADY_CONTRACT_FAR \
    "; snk-asm-contract inout %[c] %[wc] %[s1] %[s2] %[r0] %[r1] %[rbc] %[nxoff] %[anchor] %[op] %[opn] %[ns2] %[sm] %[sl]" \
    " | in %[lb] %[sx] %[kx] %[xoffB] %[yoffB] %[T0] %[limc] %[oz] %[dm] %[k8] %[arena] %[marena]" SNK_CONTRACT_BLK " %[gtb] %[ftab] %[vbm2] %[k17]\n\t"
#define SNK_STEADY_TABLE SNK_STEADY_CONTRACT SNK_STEADY_ENTER SNK_STEADY_TABLE_BODY
// (the other-case mode parks lanes inside the loop -- see SNK_STEADY_PARK_OTH -- and restores EXEC when it leaves)
#define SNK_STEADY_TABLE_OTH SNK_STEADY_CONTRACT "s_mov_b64 %[se], exec\n\t" SNK_STEADY_TABLE_BODY
#define SNK_STEADY_TABLE_BODY \
    SNK_LOOP_ALIGN \
    "1:\n\t" \
    "s_waitcnt lgkmcnt(1)\n\t"                          /* slot of cur (the slot of cur-2 may still be in flight) */ \
    "v_lshl_add_u32 v90, %[s1], 1, %[lb]\n\t" \
    "ds_read_u16 v91, v90\n\t" \
    "v_lshrrev_b32_e32 v92, 5, %[s1]\n\t" \
    "v_lshl_add_u32 v92, v92, 2, %[lb]\n\t" \
    "v_lshlrev_b32_e64 v93, %[s1], 1\n\t" \
    "ds_or_rtn_b32 v94, v92, v93 offset:1792\n\t" SNK_PADA \
    "v_add_u32_e32 v99, 0xfffe, %[c]\n\t"               /* cur - 2 in this block: the patch below, and (low 16 bits) the data of the owed put */ \
    "s_waitcnt lgkmcnt(2)\n\t" \
    "v_cndmask_b32_e64 %[s2], %[dm], %[ns2], %[sm]\n\t" /* nothing owed: the unused slot */ \
    "v_lshl_add_u32 v95, %[s2], 1, %[lb]\n\t" \
    "ds_write_b16 v95, v99\n\t" \
    "v_lshrrev_b32_e32 v97, 5, %[s2]\n\t" \
    "v_lshl_add_u32 v97, v97, 2, %[lb]\n\t" \
    "v_lshlrev_b32_e64 v98, %[s2], 1\n\t" \
    "ds_or_b32 v97, v98 offset:1792\n\t" \
    "ds_write_b16 v90, %[c]\n\t" \
    "v_cmp_eq_u32_e32 vcc, %[s2], %[s1]\n\t" \
    "v_add_u32_e32 v112, 1, %[c]\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_bfe_u32 %[t], v94, %[s1], 1\n\t" \
    "v_lshl_add_u32 %[t], %[t], 16, v91\n\t" \
    "v_cndmask_b32_e32 %[t], %[t], v99, vcc\n\t" SNK_PADB
// FAR chains (tables in global memory, u32 absolute positions).  As in the LDS loop the read of slot(cur) heads the chain
// and the case slot(cur-2) == slot(cur) is patched in by a select; the two puts are fire-and-forget stores issued AFTER
// the table load has returned (behind the candidate and refill loads): a store to a line with a read miss in flight stalls
// the CU's whole L1 pipeline until the miss returns (TCP_PENDING_STALL_CYCLES, measured: profiles/r03_far_pmc.json).
// vmcnt counts loads and stores in program order.  t is clamped to [0, 131071]: every candidate address stays inside the
// arena whatever the entry holds (a dead or stale entry gives t <= 0 -> never valid).
#ifndef SNK_FAR_LOADMOD
#define SNK_FAR_LOADMOD "sc1"      /* device scope: served by the L2, the line is not kept in the CU's L1 (which the windows of y live in) */
#endif
#define SNK_STEADY_TABLE_FAR \
    SNK_STEADY_CONTRACT_FAR \
    SNK_STEADY_ENTER \
    SNK_LOOP_ALIGN \
    "1:\n\t" \
    "s_waitcnt lgkmcnt(1)\n\t"                          /* slot of cur */ \
    "v_lshl_add_u32 v90, %[s1], 2, %[gtb]\n\t" \
    "global_load_dword v91, v90, %[ftab] " SNK_FAR_LOADMOD "\n\t" \
    "v_add_u32_e32 v96, %[vbm2], %[c]\n\t"              /* absolute cur - 2 */ \
    "v_add_u32_e32 v99, 0xfffe, %[c]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_cndmask_b32_e64 %[s2], %[dm], %[ns2], %[sm]\n\t" /* nothing owed: the unused slot */ \
    "v_lshl_add_u32 v95, %[s2], 2, %[gtb]\n\t" \
    "v_cmp_eq_u32_e32 vcc, %[s2], %[s1]\n\t" \
    "v_add_u32_e32 v97, 2, v96\n\t" \
    "v_add_u32_e32 v112, 1, %[c]\n\t" \
    "s_waitcnt vmcnt(0)\n\t" \
    "v_sub_u32_e32 %[t], v91, %[T0]\n\t" \
    "v_med3_i32 %[t], %[t], 0, %[k17]\n\t" \
    "v_cndmask_b32_e32 %[t], %[t], v99, vcc\n\t"
#if defined(SNK_FAR_DIAG) && SNK_FAR_DIAG == 1      /* diagnostic build only: no puts (wrong sizes; what the stores cost the CU) */
#define SNK_STEADY_STORES_FAR "v_mov_b32_e32 v122, v95\n\tv_mov_b32_e32 v122, v90\n\t"
#define SNK_FAR_W1 "1"
#define SNK_FAR_W0 "0"
#elif defined(SNK_FAR_DIAG) && SNK_FAR_DIAG == 2    /* diagnostic build only: the puts as atomic swaps without return */
#define SNK_STEADY_STORES_FAR \
    "global_atomic_swap v95, v96, %[ftab]\n\t" \
    "global_atomic_swap v90, v97, %[ftab]\n\t"
#define SNK_FAR_W1 "3"
#define SNK_FAR_W0 "2"
#else
#ifndef SNK_FAR_STOREMOD
#define SNK_FAR_STOREMOD ""
#endif
#define SNK_STEADY_STORES_FAR \
    "global_store_dword v95, v96, %[ftab] " SNK_FAR_STOREMOD "\n\t"          /* put(cur-2), then put(cur): same slot -> cur stays */ \
    "global_store_dword v90, v97, %[ftab] " SNK_FAR_STOREMOD "\n\t"
#define SNK_FAR_W1 "3"                                    /* the two puts are behind the loads in the count */
#define SNK_FAR_W0 "2"
#endif
// candidate window address: v104 = arena byte offset, v103 = position whose low 2 bits give the phase
#define SNK_STEADY_ADDR_DUAL \
    "v_sub_u32_e32 v100, %[t], %[sx]\n\t" \
    "v_ashrrev_i32_e32 v101, 31, v100\n\t" \
    "v_bfi_b32 v102, v101, %[xoffB], %[yoffB]\n\t" \
    "v_and_b32_e32 v101, %[kx], v101\n\t" \
    "v_add_u32_e32 v103, %[t], v101\n\t" \
    "v_lshrrev_b32_e32 v104, 2, v103\n\t" \
    "v_add_u32_e32 v104, v104, v102\n\t"
#define SNK_STEADY_ADDR_YONLY \
    "v_lshrrev_b32_e32 v104, 2, %[t]\n\t" \
    "v_add_u32_e32 v104, v104, %[yoffB]\n\t"
// shadow of the candidate load: commit of the previous probe, refill, validity (PH = phase register)
// diagnostic build only (-DSNK_PAD_LOAD=k): k redundant copies of the refill load per trip -- what one more L1 access per
// lane and trip costs (tools/gpu_padload.sh)
#ifdef SNK_PAD_LOAD
#define SNK_PADLOAD ".rept " SNK_STR(SNK_PAD_LOAD) "\n\tglobal_load_dword v122, %[nxoff], %[arena]\n\t.endr\n\t"
#define SNK_PADLOAD_W1 SNK_STR(SNK_PAD_LOAD_P1)
#define SNK_STEADY_SHADOW(PH) SNK_STEADY_SHADOW_XX(PH, "", SNK_PADLOAD)
#else
#define SNK_STEADY_SHADOW(PH) SNK_STEADY_SHADOW_X(PH, "")
#endif
#define SNK_STEADY_SHADOW_FAR(PH) SNK_STEADY_SHADOW_XX(PH, "", SNK_STEADY_STORES_FAR)
// (instantiations for sequences with exceptions: the candidate's mask window, same offset in the mask arena)
#define SNK_STEADY_MASKLOAD "global_load_dwordx2 v[118:119], v104, %[marena]\n\t"
#define SNK_STEADY_SHADOW_X(PH, LOAD2) SNK_STEADY_SHADOW_XX(PH, LOAD2, "")
#define SNK_STEADY_SHADOW_XX(PH, LOAD2, STORES) \
    "global_load_dwordx2 v[106:107], v104, %[arena]\n\t" LOAD2 \
    "v_cndmask_b32_e64 v126, 0, 4, %[sl]\n\t" \
    "v_add_u32_e32 %[nxoff], %[nxoff], v126\n\t" \
    "global_load_dword v108, %[nxoff], %[arena]\n\t" STORES SNK_PADC \
    "v_lshl_add_u32 %[rbc], v126, 2, %[rbc]\n\t" \
    "v_cndmask_b32_e64 %[op], %[op], %[opn], %[sm]\n\t" \
    "v_cndmask_b32_e64 %[anchor], %[anchor], %[c], %[sm]\n\t" \
    "v_sub_u32_e32 %[lit], %[c], %[anchor]\n\t" \
    "v_sub_u32_e32 v124, %[op], %[oz]\n\t" \
    "v_cmp_gt_u32_e64 %[sv], %[t], %[c]\n\t" \
    "v_and_b32_e32 v109, 3, " PH "\n\t" \
    "v_lshlrev_b32_e32 v109, 1, v109\n\t" \
    "v_add_u32_e32 v111, %[T0], %[t]\n\t" \
    "v_cndmask_b32_e64 v110, 0, -1, %[sv]\n\t"
#define SNK_STEADY_STRADDLE                             /* a straddling window trips the limit test */ \
    "v_cmp_gt_u32_e32 vcc, 15, v100\n\t" \
    "s_and_b64 %[ss], vcc, %[sv]\n\t" \
    "v_cndmask_b32_e64 v105, %[limc], 0, %[ss]\n\t"
// compare, next cursor, next window + LUT reads, accounting, exit test (LIM = limit register)
#ifdef SNK_PAD_LOAD
#define SNK_STEADY_REST(LIM) SNK_STEADY_REST_XX(LIM, "", SNK_STR(SNK_PAD_LOAD_P1), SNK_STR(SNK_PAD_LOAD))   /* window: 1 + k behind it; refill: k */
#else
#define SNK_STEADY_REST(LIM) SNK_STEADY_REST_X(LIM, "")
#endif
#define SNK_STEADY_REST_FAR(LIM) SNK_STEADY_REST_XX(LIM, "", SNK_FAR_W1, SNK_FAR_W0)
// (exceptions: the mask window ORed into the difference -- an exception in the candidate window ends the match there)
#define SNK_STEADY_MASKOR \
    "v_alignbit_b32 v119, v119, v118, v109\n\t" \
    "v_or_b32_e32 v113, v113, v119\n\t"
// diagnostic builds only (-DSNK_PARK=1 / 2, measured negatives of round 3): at the loop's exit test,
// lanes whose ONLY reason is their limit (1: an exception site, 2: the end of their block when the wave's lanes end their
// blocks at the same place of y) are parked -- taken out of EXEC, registers frozen in the state every lane leaves the loop
// in -- and the others walk on, instead of one wave exit and re-entry per lane.  Nothing is added to the trip itself.
#ifdef SNK_PARK
#define SNK_EC "&"
#define SNK_STEADY_ENTER "s_mov_b64 %[se], exec\n\t"
#define SNK_STEADY_PARK \
    "s_cmp_eq_u64 %[blk], 0\n\t"                       /* blk: the lanes whose limit may park them; 0 = none */ \
    "s_cbranch_scc1 .Lsnk_leave%=\n\t" \
    "v_cmp_ge_u32_e64 %[ss], %[c], %[limc]\n\t"        /* the next cursor is at the lane's real limit ... */ \
    "s_and_b64 %[ss], %[ss], %[blk]\n\t"               /* ... which is its site */ \
    "v_cmp_lt_i32_e32 vcc, 14, v125\n\t"               /* literal run, back-extension, output budget */ \
    "s_andn2_b64 %[ss], %[ss], vcc\n\t"                /* the lanes to park */ \
    "s_or_b64 vcc, vcc, %[st]\n\t" \
    "s_andn2_b64 vcc, vcc, %[ss]\n\t"                  /* a lane with another reason: leave */ \
    "s_cbranch_vccnz .Lsnk_leave%=\n\t" \
    "s_andn2_b64 exec, exec, %[ss]\n\t" \
    "s_cbranch_execnz 1b\n\t" \
    ".Lsnk_leave%=:\n\t" \
    "s_mov_b64 exec, %[se]\n\t"
#else
#define SNK_EC ""
#define SNK_STEADY_ENTER
#define SNK_STEADY_PARK
#endif
#define SNK_STEADY_REST_X(LIM, MASKOR) SNK_STEADY_REST_XX(LIM, MASKOR, "1", "0")
#define SNK_STEADY_REST_XX(LIM, MASKOR, W1, W0) SNK_STEADY_REST_XXX(LIM, MASKOR, W1, W0, "")
// (the other-case mode: the candidate's CLASS window must be 01 throughout -- XOR 0x55555555 joins the difference -- and the
// slots come from the other case's LUT, 2 KiB further up in the LDS)
#define SNK_STEADY_MASKOR_OTH \
    "v_alignbit_b32 v119, v119, v118, v109\n\t" \
    "v_xor_b32_e32 v119, 0x55555555, v119\n\t" \
    "v_or_b32_e32 v113, v113, v119\n\t"
// In the other-case mode the lanes of a wave walk the same stretch a few trips apart and each ends it at its own trip: a lane
// whose ONLY reason to leave is its limit (the stretch's end, or the block's) is taken out of EXEC -- registers frozen in the
// state every lane leaves the loop in -- and the others walk on; the wave leaves when a lane has another reason or none is
// left.  (One wave exit per lane at the end of every stretch cost more than the stretch itself: 164 k cycles for ~95 trips.)
#define SNK_STEADY_PARK_OTH \
    "v_cmp_ge_u32_e64 %[ss], %[c], %[limc]\n\t"      /* the next cursor is at the lane's real limit */ \
    "v_cmp_lt_i32_e32 vcc, 14, v125\n\t"             /* literal run, back-extension, output budget */ \
    "s_andn2_b64 %[ss], %[ss], vcc\n\t"              /* the lanes to park */ \
    "s_or_b64 vcc, vcc, %[st]\n\t" \
    "s_andn2_b64 vcc, vcc, %[ss]\n\t"                /* a lane with another reason: leave */ \
    "s_cbranch_vccnz .Lsnk_oleave%=\n\t" \
    "s_andn2_b64 exec, exec, %[ss]\n\t" \
    "s_cbranch_execnz 1b\n\t" \
    ".Lsnk_oleave%=:\n\t" \
    "s_mov_b64 exec, %[se]\n\t"
#define SNK_STEADY_REST_OTH(LIM) SNK_STEADY_REST_CORE(LIM, SNK_STEADY_MASKOR_OTH, "1", "0", " offset:2048", SNK_STEADY_PARK_OTH)
#define SNK_STEADY_REST_XXX(LIM, MASKOR, W1, W0, LUTOFF) SNK_STEADY_REST_CORE(LIM, MASKOR, W1, W0, LUTOFF, SNK_STEADY_PARK)
#define SNK_STEADY_REST_CORE(LIM, MASKOR, W1, W0, LUTOFF, PARK) \
    "s_waitcnt vmcnt(" W1 ")\n\t" SNK_PADD \
    "v_alignbit_b32 v113, v107, v106, v109\n\t" \
    "v_xor_b32_e32 v113, v113, %[wc]\n\t" MASKOR \
    "v_lshrrev_b32_e32 v114, 8, v113\n\t" \
    "v_ffbl_b32_e32 v114, v114\n\t" \
    "v_and_b32_e32 v114, v114, v110\n\t" \
    "v_cmp_lt_u32_e64 %[sm], 7, v114\n\t" \
    "v_lshrrev_b32_e32 v115, 1, v114\n\t" \
    "v_add_u32_e32 v115, v115, %[c]\n\t" \
    "v_cndmask_b32_e64 %[c], v112, v115, %[sm]\n\t" \
    "v_sub_u32_e32 v116, %[c], %[rbc]\n\t" \
    "v_cmp_lt_u32_e64 %[sl], 15, v116\n\t" \
    "v_lshlrev_b32_e32 v116, 1, v116\n\t" \
    "v_lshl_or_b32 v120, v113, 24, %[k8]\n\t" \
    "s_waitcnt vmcnt(" W0 ")\n\t" \
    "v_cndmask_b32_e64 %[r0], %[r0], %[r1], %[sl]\n\t" \
    "v_cndmask_b32_e64 %[r1], %[r1], v108, %[sl]\n\t" \
    "v_alignbit_b32 %[wc], %[r1], %[r0], v116\n\t" \
    "v_lshrrev_b32_e32 v117, 7, %[wc]\n\t" \
    "v_and_b32_e32 v117, 0x7fe, v117\n\t" \
    "ds_read_u16 %[s1], v117" LUTOFF "\n\t" \
    "v_lshrrev_b32_e32 v118, 3, %[wc]\n\t" \
    "v_and_b32_e32 v118, 0x7fe, v118\n\t" \
    "ds_read_u16 %[ns2], v118" LUTOFF "\n\t" SNK_PADE \
    "v_ffbh_u32_e32 v120, v120\n\t" \
    "v_lshrrev_b32_e32 v120, 1, v120\n\t" \
    "v_min3_u32 v120, v120, %[lit], v111\n\t" \
    "v_sub_u32_e32 v121, %[lit], v120\n\t" \
    "v_add3_u32 %[opn], %[op], v121, 3\n\t" \
    "v_add_u32_e32 v123, 11, v120\n\t" \
    "v_max3_i32 v125, v123, %[lit], v124\n\t" \
    "v_cmp_lt_i32_e32 vcc, 14, v125\n\t" \
    "v_cmp_ge_u32_e64 %[st], %[c], " LIM "\n\t" \
    "s_or_b64 vcc, vcc, %[st]\n\t" \
    "s_cbranch_vccz 1b\n\t" \
    PARK \
    "s_waitcnt lgkmcnt(0)\n\t"
#define SNK_STEADY_OPERANDS SNK_STEADY_OPERANDS_X()
#define SNK_STEADY_OPERANDS_FAR SNK_STEADY_OPERANDS_X(, [gtb] "v"(gtb), [ftab] "s"(ftab), [vbm2] "v"(vb - 2u), [k17] "s"(131071))
#define SNK_STEADY_OPERANDS_X(...) \
    /* (SNK_EC: under -DSNK_PARK the read-write operands are early-clobber -- the input-only blk = 0 was otherwise given the */ \
    /* register pair of sl = 0 and overwritten by the loop.  The shipped loop has no input-only operand of provably equal */ \
    /* value and keeps the plain form: with early-clobber the register numbering changes and the loop runs 2 % slower, */ \
    /* 448 k against 457 k pair-compr./s -- operand banks.) */ \
    : [c] "+" SNK_EC "v"(c), [wc] "+" SNK_EC "v"(wc), [s1] "+" SNK_EC "v"(s1), [s2] "+" SNK_EC "v"(s2), [r0] "+" SNK_EC "v"(r0), [r1] "+" SNK_EC "v"(r1), \
      [rbc] "+" SNK_EC "v"(rbc), [nxoff] "+" SNK_EC "v"(nxoff), [anchor] "+" SNK_EC "v"(anchor_c), [op] "+" SNK_EC "v"(op), \
      [opn] "+" SNK_EC "v"(opn), [ns2] "+" SNK_EC "v"(ns2), [sm] "+" SNK_EC "s"(sm), [sl] "+" SNK_EC "s"(sl), \
      [t] "=&v"(t), [lit] "=&v"(lit), [sv] "=&s"(sv), [ss] "=&s"(ss), [st] "=&s"(st), [se] "=&s"(se) \
    : [lb] "v"(lds_off), [sx] "v"(sx), [kx] "v"(kx), [xoffB] "v"(xoffB), [yoffB] "v"(yoffB), \
      [T0] "v"(T0), [limc] "v"(limc), [oz] "v"(oz), [dm] "v"(dm), [k8] "s"(0x00800000u), \
      [arena] "s"(arena), [marena] "s"(marena) SNK_OPERAND_BLK __VA_ARGS__ \
    : "memory", "vcc", "scc", \
      "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", \
      "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", \
      "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

// A lane may run the steady loop when its next probe is an ordinary one: inside an open block
// that has not bailed out, search step 1, literal run and output budget far from their rare
// ranges, cursor served by the reservoir's source.
#ifndef SNK_EXC_WAITMAX
#define SNK_EXC_WAITMAX 64u      // (exceptions) ... or have waited this many rounds of the wave loop
#endif
#ifndef SNK_EXC_GATHER
#define SNK_EXC_GATHER    16384u   // (exceptions) lanes waiting at a site are served once no running lane is fewer bases behind them
#endif
#define SNK_FAST_MAXLIT   15u      // the steady loop leaves at literal runs >= 15 (length-extension bytes)
#define SNK_FAST_ZONE     80u      // ... and when op comes within 80 bytes of the block's output budget
template <bool EXC>
__device__ __forceinline__ bool snk_fast_eligible(const SnkFastLane &L)
{
    const uint32_t o = L.cur - 4u - L.w.rb;
    return (L.cur + L.step <= L.mfl1) & (L.step == 1u) & (L.nb < 63u + SNK_FAST_MAXLIT) &
           (L.op + SNK_FAST_ZONE <= L.olimit) & (o < 32u) & (L.cur <= L.w.lim) & (!EXC || L.cur < L.xlim);
}

// The steady loop: one probe per trip for every lane of the wave, no divergent branch, ONE
// wave-uniform exit (ballot) taken when any lane needs service; the lanes are then all handed to
// snk_fast_finish with the state "table operations of the probe done, match not evaluated".
//
// Coordinates.  Everything is relative to the block's virtual base vb = L.base (block start minus
// k3, k3 = -lx mod 4): c = cur - vb is what the table stores; a candidate is t = e + 65536*hit,
// the offset from vb - 65536 (hit = "written in this block").  k3 makes (vb - lx - 4) a multiple
// of 4, so a candidate window inside y starts at arena byte yoffB + (t >> 2), bit 2*(t & 3), with
// no per-probe phase arithmetic; inside x the phase kx is added.  Windows straddling the x/y
// seam (15 values of t) leave through the exit.  All loads stay inside the arena for ANY t
// (16 KiB of slack before the first and after the last sequence), so validity is applied to the
// compare result, not to the address.
//
// Order of the table operations: liblz4 does put(cur-2), get(cur), put(cur).  Here the read of
// slot(cur) is issued first (it heads the dependent chain) and the case slot(cur-2) == slot(cur)
// is patched in by one select; the bitmap word comes back from the OR that sets the bit of cur
// (ds_or_rtn_b32).
//
// ASM = true: the loop proper is the hand-scheduled gfx950 code below (same dataflow, statement for
// statement); ASM = false: the C++ statement of it, which is also what the CPU emulation runs.
// OTH (with EXC, C++ statement only): the other-case mode, see snk_exc_other_ready; okey = lut_okey.
template <bool ASM, bool EXC, bool FAR, bool OTH = false>
__device__ __forceinline__ void snk_fast_steady(SnkFastLane &L, snk_g8 *const arena, snk_g8 *const marena, uint16_t *tbl, uint32_t *bm,
                                                uint32_t *gt, SNK_AS1 uint32_t *const ftab, uint32_t gtb,
                                                const uint16_t *slot, uint32_t lds_off, uint32_t round_bases,
                                                const uint16_t *okey SNK_PROF_ARG)
{
    static_assert(!OTH || (EXC && !FAR), "the other-case mode: sequences with exceptions, tables in LDS");
    SnkWin &w = L.w;                              // arena: the kernel argument (wave-uniform: the asm addresses it through SGPRs)
#ifdef SNK_STATS
    const unsigned long long stat_te = clock64();
    const bool stat_first = (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
    (void)stat_first;
#endif
    // OTH: the chain's LDS region holds its OTHER-CASE table for the length of the stretch (snk_oth_swap_in), keyed by the other
    // case's LUT, which the kernels for sequences with exceptions keep 2 KiB further up in the LDS
#ifdef SNK_HOST_EMU
    const uint16_t *const lut0 = slot + (OTH ? 1024 : 0);
#else
    (void)slot;                                   // the LUT sits at LDS address 0 (host checks: no static LDS)
    const SNK_AS3 uint16_t *const lut0 = (const SNK_AS3 uint16_t *)0 + (OTH ? 1024 : 0);
#endif
    (void)okey;
    const uint32_t vb = L.base;
    const int32_t T0 = (int32_t)(vb - 65536u);                    // stream position of t = 0
    const int32_t X0 = T0 - 4, Y0 = T0 - 4 - (int32_t)L.s.lx;    // source index of the window of t = 0
    const uint32_t kx = (uint32_t)X0 & 3u;                        // (Y0 & 3) == 0 by the choice of k3
    const uint32_t xoffB = L.s.xoff + (uint32_t)(X0 >> 2), yoffB = L.s.yoff + (uint32_t)(Y0 >> 2);
    const int32_t sx = (int32_t)L.s.lx - 11 - T0;                 // t < sx: window inside x; t >= sx + 15: inside y
    const uint32_t limw = w.lim == 0xFFFFFFFFu ? w.lim : w.lim + 1u;
    uint32_t lim_abs = L.mfl1 < limw ? L.mfl1 : limw;
    if (EXC) { const uint32_t xl = OTH ? L.olim : L.xlim; lim_abs = lim_abs < xl ? lim_abs : xl; }   // ... the cursor window must stay clean (OTH: of class 01)
    if (EXC && round_bases != 0xFFFFFFFFu && lim_abs - L.cur > round_bases) lim_abs = L.cur + round_bases;   // a short round
    const uint32_t limc = lim_abs - vb;                           // next probe position >= limc: service
    // (EXC, wave-uniform) can a lane of this run read an entry whose window holds an exception?  Only then the mask window is loaded.
    const bool need_mask = OTH || (EXC && __any(L.cur < L.mask_until));
    const uint32_t kcls = OTH ? 0x55555555u : 0u;                 // the class every base of the candidate must have
    const int32_t olimZ = (int32_t)L.olimit - (int32_t)SNK_FAST_ZONE + 10;      // olimit - 70: eligibility needs op <= olimit - 80

    uint32_t c = L.cur - vb, anchor_c = L.anchor - vb, op = L.op;
    uint32_t r0 = w.r0, r1 = w.r1, r2 = w.nx;
    uint32_t rbc = w.rb + 4u - vb;                                // c - rbc = offset of the cursor window in (r0, r1, r2)
    uint32_t nxoff = w.soff + ((w.rb + 32u - w.org) >> 2);        // arena offset of the bases [rb+32, rb+48)
    uint32_t wc, s1, s2;
    {   // window and slots of the first probe (slides the reservoir if the cursor has left r0)
        const uint32_t no = c - rbc;
        const bool sl = no >= 16u;
        r0 = sl ? r1 : r0; r1 = sl ? r2 : r1;
        rbc += sl ? 16u : 0u; nxoff += sl ? 4u : 0u;
        wc = __builtin_amdgcn_alignbit(r1, r0, 2u * no);
        s1 = lut0[(wc >> 8) & 1023u];
        s2 = L.pending ? (uint32_t)lut0[(wc >> 4) & 1023u] : (SNK_FSLOTS - 1u);     // nothing owed: the unused slot
    }
    uint32_t t; bool valid;
#ifdef SNK_STATS
    unsigned int stat_otrips = 0u;
    const unsigned long long stat_t0 = clock64();
    P.prologue += stat_t0 - stat_te;
#endif
#ifndef SNK_HOST_EMU
    if (ASM) {
        uint32_t lit;
        uint64_t sv, ss, st, se;                                      // wave masks: valid, straddle, limit; EXEC at entry
        // (EXC) the lanes whose limit is an exception site: parked at the loop's exit test instead of forcing a wave exit each
        uint64_t blk = 0ull;
#if defined(SNK_PARK) && SNK_PARK == 1          /* diagnostic: park at exception sites */
        if (EXC) blk = __builtin_amdgcn_ballot_w64(lim_abs == L.xlim);
#elif defined(SNK_PARK) && SNK_PARK == 2        /* diagnostic: park at block ends when the wave's lanes end their blocks at the same place of y */
        if (!EXC && !FAR && L.cur >= L.s.lx &&
            __all((L.mfl1 - L.s.lx) == (uint32_t)__builtin_amdgcn_readfirstlane((int)(L.mfl1 - L.s.lx)) && L.cur >= L.s.lx))
            blk = __builtin_amdgcn_ballot_w64(lim_abs == L.mfl1);
#endif
        blk = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(blk >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)blk);                      // (an SGPR pair for the asm, whatever the compiler thinks of it)
        uint64_t sm = __builtin_amdgcn_ballot_w64(L.pending);         // match of the previous probe (its commit is owed)
        uint64_t sl = 0;                                              // slide of the previous probe (none owed)
        uint32_t opn = op;                                            // ... committing it again changes nothing
        uint32_t ns2 = lut0[(wc >> 4) & 1023u];
        const uint32_t oz = (uint32_t)(olimZ - 14), dm = SNK_FSLOTS - 1u;
        // every lane's block wholly > 64 KiB past its seam (no t can land in x or on the seam)?
        if (FAR) {
            if (__all(sx + 15 <= 0))
                asm volatile(SNK_STEADY_TABLE_FAR SNK_STEADY_ADDR_YONLY SNK_STEADY_SHADOW_FAR("%[t]") SNK_STEADY_REST_FAR("%[limc]")
                             SNK_STEADY_OPERANDS_FAR);
            else
                asm volatile(SNK_STEADY_TABLE_FAR SNK_STEADY_ADDR_DUAL SNK_STEADY_SHADOW_FAR("v103") SNK_STEADY_STRADDLE SNK_STEADY_REST_FAR("v105")
                             SNK_STEADY_OPERANDS_FAR);
        } else if (OTH) {
            if (__all(sx + 15 <= 0))
                asm volatile(SNK_STEADY_TABLE_OTH SNK_STEADY_ADDR_YONLY SNK_STEADY_SHADOW_X("%[t]", SNK_STEADY_MASKLOAD)
                             SNK_STEADY_REST_OTH("%[limc]") SNK_STEADY_OPERANDS);
            else
                asm volatile(SNK_STEADY_TABLE_OTH SNK_STEADY_ADDR_DUAL SNK_STEADY_SHADOW_X("v103", SNK_STEADY_MASKLOAD) SNK_STEADY_STRADDLE
                             SNK_STEADY_REST_OTH("v105") SNK_STEADY_OPERANDS);
        } else if (!EXC || !need_mask) {
            if (__all(sx + 15 <= 0))
                asm volatile(SNK_STEADY_TABLE SNK_STEADY_ADDR_YONLY SNK_STEADY_SHADOW("%[t]") SNK_STEADY_REST("%[limc]")
                             SNK_STEADY_OPERANDS);
            else
                asm volatile(SNK_STEADY_TABLE SNK_STEADY_ADDR_DUAL SNK_STEADY_SHADOW("v103") SNK_STEADY_STRADDLE SNK_STEADY_REST("v105")
                             SNK_STEADY_OPERANDS);
        } else {
            if (__all(sx + 15 <= 0))
                asm volatile(SNK_STEADY_TABLE SNK_STEADY_ADDR_YONLY SNK_STEADY_SHADOW_X("%[t]", SNK_STEADY_MASKLOAD)
                             SNK_STEADY_REST_X("%[limc]", SNK_STEADY_MASKOR) SNK_STEADY_OPERANDS);
            else
                asm volatile(SNK_STEADY_TABLE SNK_STEADY_ADDR_DUAL SNK_STEADY_SHADOW_X("v103", SNK_STEADY_MASKLOAD) SNK_STEADY_STRADDLE
                             SNK_STEADY_REST_X("v105", SNK_STEADY_MASKOR) SNK_STEADY_OPERANDS);
        }
        c = lit + anchor_c;                                           // the loop keeps the NEXT cursor in c
        valid = t > c;
    } else
#endif
    for (;;) {
        // ---- table: read slot(cur) + its bitmap word (the OR returns the old word), owed put, put(cur) ----
#ifdef SNK_HOST_EMU
        if (OTH) snk_emu_oth_trips++;
#endif
#ifdef SNK_STATS
        if (OTH) stat_otrips++;
#endif
        if (FAR) {                                                // absolute positions in global memory, liblz4's order
            gt[s2] = vb + c - 2u;
            const int32_t ts = (int32_t)(gt[s1] - (uint32_t)T0);
            gt[s1] = vb + c;
            t = ts < 0 ? 0u : (ts > 131071 ? 131071u : (uint32_t)ts);
        } else {
            const uint32_t bit1 = s1 & 31u;
            const uint32_t e = tbl[s1];
            const uint32_t bw = atomicOr(&bm[s1 >> 5], 1u << bit1);
            tbl[s2] = (uint16_t)(c - 2u);
            atomicOr(&bm[s2 >> 5], 1u << (s2 & 31u));
            tbl[s1] = (uint16_t)c;
            t = e + (((bw >> bit1) & 1u) << 16);
            t = (s2 == s1) ? 65534u + c : t;                      // the owed put went to the same slot: candidate cur-2
        }
        valid = t > c;                                            // this block: t >= 65536 > c; previous block: e > c

        // ---- candidate window (global, L1) and the reservoir refill, in flight together ----
        const bool inx = (int32_t)t < sx;
        const uint32_t tt = t + (inx ? kx : 0u);
        const uint64_t v = snk_ld8g(arena + (size_t)((inx ? xoffB : yoffB) + (tt >> 2)));
        r2 = snk_ld4g(arena + (size_t)nxoff);
        const bool straddle = valid & ((uint32_t)((int32_t)t - sx) < 15u);
        const uint32_t lit = c - anchor_c;
        const uint32_t wd = __builtin_amdgcn_alignbit((uint32_t)(v >> 32), (uint32_t)v, 2u * (tt & 3u));
        uint32_t wm = 0u;                                         // (EXC) the candidate's mask window: 11 where a byte is not one of ACGT
        if (EXC && need_mask) {
            const uint64_t mv = snk_ld8g(marena + (size_t)((inx ? xoffB : yoffB) + (tt >> 2)));
            wm = __builtin_amdgcn_alignbit((uint32_t)(mv >> 32), (uint32_t)mv, 2u * (tt & 3u));
        }

        // ---- compare, next cursor ----
        const uint32_t x = (wc ^ wd) | (wm ^ kcls);
        const uint32_t r = snk_ffbl(x >> 8);                      // 2 * equal bases from cur; 0xFFFFFFFF: all 12
        const bool m = valid & (r >= 8u);
        const uint32_t e2 = c + (r >> 1);                         // all 12 equal: huge; past the match limit: >= limc -> service
        const uint32_t ncur = m ? e2 : c + 1u;

        // ---- window and slot LUT reads of the next probe ----
        const uint32_t no = ncur - rbc;                           // 0..27
        const bool sl = no >= 16u;
        const uint32_t lo = sl ? r1 : r0, hi = sl ? r2 : r1;
        const uint32_t nwc = __builtin_amdgcn_alignbit(hi, lo, 2u * no);
        const uint32_t ns1 = lut0[(nwc >> 8) & 1023u];
        const uint32_t ns2 = lut0[(nwc >> 4) & 1023u];

        // ---- this probe's accounting, in the shadow of the LUT reads ----
        const uint32_t eq = (uint32_t)__builtin_clz(((x & 0xFFu) << 24) | 0x00800000u) >> 1;   // equal bases before cur, 0..4
        uint32_t b = eq < lit ? eq : lit;
        { const uint32_t cand = (uint32_t)(T0 + (int32_t)t); b = b < cand ? b : cand; }          // not before the stream start
        const uint32_t opn = op + (lit - b) + 3u;                 // token + literals + offset, no extension bytes
        // service: literal run >= 15, back-extension reaches 4 (may go on), output budget near
        int32_t mx = (int32_t)(b + 11u) > (int32_t)lit ? (int32_t)(b + 11u) : (int32_t)lit;
        { const int32_t z = (int32_t)op - olimZ + 14; mx = mx > z ? mx : z; }       // op before this probe > olimit - 70
        const bool svc = (mx >= 15) | (ncur >= limc) | straddle;
        SNK_TRACE_REC(3u, vb + c, (uint32_t)(T0 + (int32_t)t), (r << 24) | (m ? 0x800000u : 0u) | (valid ? 0x400000u : 0u) | (e2 & 0x3FFFFFu), vb + c);
#ifdef SNK_STATS
        if (!OTH && stat_first) SNK_COUNT(15);                    // (the other-case mode keeps its account undistorted: no atomics per trip)
        if (!OTH && svc) {          // why lanes ask for service (a lane may have several reasons; an exit may serve several lanes)
            if ((int32_t)lit >= 15) SNK_COUNT(16);
            if (b >= 4u) SNK_COUNT(17);
            if ((int32_t)op - olimZ + 14 >= 15) SNK_COUNT(18);
            if (r == 0xFFFFFFFFu && valid) SNK_COUNT(19);                 // 12 equal bases forward
            else if (ncur >= limc) { if (vb + ncur >= L.mfl1) SNK_COUNT(20); else SNK_COUNT(21); }      // block end / another limit
            if (straddle) SNK_COUNT(22);
        }
#endif
        // one wave-uniform exit
        if (__builtin_amdgcn_ballot_w64(svc) != 0ull) break;

        // ---- commit ----
        op = m ? opn : op; anchor_c = m ? ncur : anchor_c;
        r0 = lo; r1 = hi; rbc += sl ? 16u : 0u; nxoff += sl ? 4u : 0u;
        c = ncur; wc = nwc; s1 = ns1; s2 = m ? ns2 : (SNK_FSLOTS - 1u);
    }
    SNK_COUNT_HEAVY(0);
#ifdef SNK_STATS
    const unsigned long long stat_t1 = clock64();
    P.loop += stat_t1 - stat_t0; P.entries++;
    if (OTH) {      // wave trips of this run = the most any lane made; lane-trips = their sum (every lane keeps the wave's sums)
        unsigned int mx = stat_otrips, sm = stat_otrips;
        for (int o = 32; o; o >>= 1) { const unsigned int v = (unsigned int)__shfl_xor((int)mx, o), w2 = (unsigned int)__shfl_xor((int)sm, o); mx = v > mx ? v : mx; sm += w2; }
        P.otrips += mx; P.olanes += sm; P.oruns++;
    }
#endif
    // hand every lane over in the state "table operations of the probe at c done, match not evaluated"
    L.cur = vb + c; L.anchor = vb + anchor_c; L.op = op; L.step = 1u; L.nb = 63u + (c - anchor_c);
    w.rb = 0x80000000u;                                           // reservoir not kept: the head re-seats it
    if (OTH) {
        const uint32_t cur = vb + c, cand = (uint32_t)(T0 + (int32_t)t);
        if (!snk_exc_finish_win(L, cur, cand, valid, snk_fetch32(L.s, cur), snk_fetch32m(L.s, cur)))
            snk_exc_finish(L, cur, cand, valid, snk_ld8(L.g, cur));
    } else if (EXC) {
        snk_fast_exc_handover(L, vb + c, (uint32_t)(T0 + (int32_t)t), valid);
    } else {
        snk_fast_finish(L, vb + c, (uint32_t)(T0 + (int32_t)t), valid);
    }
#ifdef SNK_STATS
    P.finish += clock64() - stat_t1;
#endif
}

// ---- the speculative loop as gfx950 code (snk_fast_steady_spec<true>) ----------------------------------------------
// Same dataflow as the C++ statement below, scheduled like the loop above.  Differences from that statement, all exact:
//  * role 1 reads the table with role 0 at the top; role 0's two puts of the same trip are patched into role 1's result by
//    selects (partner's slots through one DPP move), as the owed put is for everybody;
//  * role 1's two puts are issued in the trip its probe counted in, as soon as that is known (after role 0's, liblz4's
//    order), with dummy addresses when it did not count;
//  * "counts" (com) = role 0's next cursor equals role 1's cursor; role 0 withholds its next cursor (-1) when it is sure to
//    need service, so com implies that role 0's probe is an ordinary 5-base match; a counted role-1 probe that needs service
//    itself (limit, straddling candidate, 12 equal bases) ends the loop and is handed over in role 0's place;
//  * the chain's accounting is committed one trip late, in the shadow of the candidate load (masks sq / scm / sc0).
#define SNK_SPEC_DEF_X(VAL) \
    "s_and_b64 exec, %[sc], %[r1m]\n\t"                 /* role 1's puts, if its probe counted: the other lanes sit the four operations out */ \
    "ds_write_b16 v95, v127\n\t"                        /* (issued and counted all the same: the waits at the top see four operations every trip) */ \
    "ds_or_b32 v97, v98 offset:1792\n\t" \
    "ds_write_b16 v90, " VAL "\n\t" \
    "ds_or_b32 v92, v93 offset:1792\n\t" \
    "s_mov_b64 exec, %[ex]\n\t"
// (measured the same within the noise between boxes, ~0.5 %: these puts before the cursor update instead -- one instruction
// fewer -- and one or two waits for the two slots at the top)
#define SNK_SPEC_DEF_A
#define SNK_SPEC_DEF_B SNK_SPEC_DEF_X("v104")
#define SNK_SPEC_W0 "4"
#define SNK_SPEC_W1
#define SNK_SPEC_CUR "v_add_u32_e32 v104, 2, v127\n\t"      /* this trip's cursor (+ 65536: 16-bit data of role 1's put) */
// what depends on the cursor alone is computed as soon as the cursor is known, at the loop's bottom where the wave waits
// for the LUT anyway (and once in front of the loop): cur + 1; cur - 2 in this block (bit 16 set: the patch value of an owed
// put, and with its low 16 bits the data of put(cur-2)); for role 1 the partner's put(cur) = this cursor - 5 and its put(cur-2)
#define SNK_SPEC_CURS \
    "v_add_u32_e32 v112, 1, %[c]\n\t" \
    "v_add_u32_e32 v102, 0xfffb, %[c]\n\t" \
    "v_add_u32_e32 v103, 0xfff9, %[c]\n\t" \
    "v_add_u32_e32 v127, 0xfffe, %[c]\n\t"
#define SNK_SPEC_CONTRACT \
    "; snk-asm-contract inout %[c] %[wc] %[s1] %[s2] %[r0] %[r1] %[rbc] %[nxoff] %[anchor] %[op] %[opn] %[ns2] %[sm] %[sl] %[scm] %[sc0] %[sq] %[sc]" \
    " | in %[lb] %[five] %[fivec] %[sx] %[kx] %[xoffB] %[yoffB] %[T0] %[limc] %[oz] %[dm] %[k8] %[arena] %[marena] %[r1m] %[vz]\n\t"
#define SNK_SPEC_TABLE \
    SNK_SPEC_CONTRACT \
    "s_mov_b64 %[ex], exec\n\t"                         /* the lanes of the loop */ \
    SNK_SPEC_CURS \
    SNK_LOOP_ALIGN \
    "1:\n\t" \
    "s_waitcnt lgkmcnt(" SNK_SPEC_W0 ")\n\t"            /* the slots of cur and cur-2 (behind them: role 1's four put operations) */ \
    "v_lshl_add_u32 v90, %[s1], 1, %[lb]\n\t" \
    "ds_read_u16 v91, v90\n\t" \
    "v_lshrrev_b32_e32 v92, 5, %[s1]\n\t" \
    "v_lshl_add_u32 v92, v92, 2, %[lb]\n\t" \
    "v_lshlrev_b32_e64 v93, %[s1], 1\n\t" \
    "v_cndmask_b32_e64 v99, v93, 0, %[r1m]\n\t"         /* role 1 only reads */ \
    "ds_or_rtn_b32 v94, v92, v99 offset:1792\n\t" SNK_PADA \
    SNK_SPEC_W1 \
    "v_cndmask_b32_e64 %[s2], %[dm], %[ns2], %[sm]\n\t" /* nothing owed: the unused slot (role 1: always owed) */ \
    "v_lshl_or_b32 v100, %[s2], 16, %[s1]\n\t"          /* both slots, for the partner */ \
    "v_cmp_eq_u32_e32 vcc, %[s2], %[s1]\n\t" \
    "v_lshl_add_u32 v95, %[s2], 1, %[lb]\n\t"           /* (two instructions between the write of v100 and its DPP read) */ \
    "v_mov_b32_dpp v101, v100 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_cmp_eq_u16_sdwa %[ss], %[s1], v101 src0_sel:WORD_0 src1_sel:WORD_1\n\t" \
    "v_cmp_eq_u16_sdwa %[st], %[s1], v101 src0_sel:WORD_0 src1_sel:WORD_0\n\t" \
    "s_and_b64 %[ss], %[ss], %[r1m]\n\t" \
    "s_and_b64 %[st], %[st], %[r1m]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_bfe_u32 %[t], v94, %[s1], 1\n\t" \
    "v_lshl_add_u32 %[t], %[t], 16, v91\n\t" \
    "v_cndmask_b32_e64 %[t], %[t], v103, %[ss]\n\t" \
    "v_cndmask_b32_e64 %[t], %[t], v102, %[st]\n\t" \
    "v_cndmask_b32_e32 %[t], %[t], v127, vcc\n\t" SNK_PADB
// (sets with exceptions, when a lane of the run may read an entry whose window holds one: the candidate's class window,
// same offset in the class arena, ORed into the difference -- as in the one-lane loop)
#define SNK_SPEC_MASKLOAD "global_load_dwordx2 v[88:89], v104, %[marena]\n\t"
#define SNK_SPEC_MASKOR \
    "v_alignbit_b32 v89, v89, v88, v109\n\t" \
    "v_or_b32_e32 v113, v113, v89\n\t"
// opn = the output position after this lane's probe if it is a match: token + literals - back-extension + offset
#define SNK_SPEC_OPN \
    "v_min3_u32 v120, v120, %[lit], v111\n\t"           /* back-extension: not beyond the literals, not before the stream start */ \
    "v_sub_u32_e32 v121, %[lit], v120\n\t" \
    "v_add3_u32 %[opn], %[op], v121, 3\n\t"
#define SNK_SPEC_SHADOW(PH) SNK_SPEC_SHADOW_X(PH, "")
#define SNK_SPEC_SHADOW_X(PH, LOAD2) \
    "global_load_dwordx2 v[106:107], v104, %[arena]\n\t" LOAD2 \
    "v_cndmask_b32_e64 v126, 0, 4, %[sl]\n\t" \
    "v_add_u32_e32 %[nxoff], %[nxoff], v126\n\t" \
    "global_load_dword v108, %[nxoff], %[arena]\n\t" SNK_PADC \
    SNK_SPEC_OPN                                                /* the last trip's account, here where the wave waits anyway */ \
    "s_and_b64 %[scm], %[sc], %[sm]\n\t"                        /* (sm holds role 1's bits too: role 1's copies are not used) */ \
    "s_andn2_b64 %[sc0], %[sc], %[sm]\n\t" \
    "v_lshrrev_b32_e32 v97, 5, %[s2]\n\t"                       /* role 0: put(cur-2), put(cur) -- behind the read, patched in for role 1 */ \
    "v_lshl_add_u32 v97, v97, 2, %[lb]\n\t" \
    "v_lshlrev_b32_e64 v98, %[s2], 1\n\t" \
    "s_andn2_b64 exec, exec, %[r1m]\n\t" \
    "ds_write_b16 v95, v127\n\t" \
    "ds_or_b32 v97, v98 offset:1792\n\t" \
    "ds_write_b16 v90, %[c]\n\t" \
    "s_or_b64 exec, exec, %[r1m]\n\t" \
    "v_lshl_add_u32 %[rbc], v126, 2, %[rbc]\n\t" \
    "v_cndmask_b32_e64 %[op], %[op], %[opn], %[sq]\n\t"         /* last trip: this lane's probe was a match (role 1 keeps no account: its op only drifts) */ \
    "v_add_u32_e32 v109, 3, %[op]\n\t" \
    "v_add_u32_e32 v110, -1, %[c]\n\t" \
    "v_cndmask_b32_e64 %[op], %[op], v109, %[scm]\n\t"          /* ... role 1's counted and was one: token + offset */ \
    "v_cndmask_b32_e64 %[anchor], %[anchor], v110, %[sc0]\n\t"  /* ... counted and was none: the anchor is its cursor */ \
    "v_cndmask_b32_e64 %[anchor], %[anchor], %[c], %[sm]\n\t"   /* after a match (role 1: always) the anchor is the cursor */ \
    "v_sub_u32_e32 %[lit], %[c], %[anchor]\n\t" \
    "v_sub_u32_e32 v124, %[op], %[oz]\n\t" \
    "v_cmp_gt_u32_e64 %[sv], %[t], %[c]\n\t" \
    "v_and_b32_e32 v109, 3, " PH "\n\t" \
    "v_lshlrev_b32_e32 v109, 1, v109\n\t" \
    "v_add_u32_e32 v111, %[T0], %[t]\n\t" \
    "v_cndmask_b32_e64 v110, 0, -1, %[sv]\n\t"
// (after the straddle test of the DUAL form) what is known before the compare: Y = the cursor role 1 probes at (role 0: cur+5),
// never matched (-2) when it is at the limit; role 0 will need service for its literal run, the budget (sp) or -- unless the
// candidate lies within 4 bases of the stream start -- a back-extension of 4 (sb & the compare's low byte); a lane that
// withholds its cursor leaves the loop in any case
#define SNK_SPEC_PRE(STRAD_OR) \
    "v_add_u32_e32 v119, %[c], %[fivec]\n\t" \
    "v_max_i32_e32 v122, %[lit], v124\n\t" \
    "v_cmp_ge_u32_e64 %[st], v119, %[limc]\n\t" \
    "v_cmp_lt_i32_e64 %[sp], 14, v122\n\t" \
    "v_cmp_lt_u32_e64 %[sb], 3, %[lit]\n\t" \
    "v_cndmask_b32_e64 v119, v119, -2, %[st]\n\t" \
    "v_sub_u32_e32 v96, v112, %[five]\n\t"              /* role 0's cursor + 1 */ \
    SNK_SPEC_CUR \
    STRAD_OR
#define SNK_SPEC_REST(LIM) SNK_SPEC_REST_X(LIM, "", "1")
#define SNK_SPEC_REST_X(LIM, MASKOR, W1) \
    "s_waitcnt vmcnt(" W1 ")\n\t" SNK_PADD \
    "v_alignbit_b32 v113, v107, v106, v109\n\t" \
    "v_xor_b32_e32 v113, v113, %[wc]\n\t" MASKOR \
    "v_lshrrev_b32_e32 v114, 8, v113\n\t" \
    "v_cmp_eq_u32_sdwa vcc, v113, %[vz] src0_sel:BYTE_0 src1_sel:DWORD\n\t"   /* the 4 bases before cur are equal too */ \
    "v_ffbl_b32_e32 v114, v114\n\t" \
    "v_and_b32_e32 v114, v114, v110\n\t" \
    "v_cmp_lt_u32_e64 %[sq], 7, v114\n\t"               /* this lane's probe is a match */ \
    "v_lshrrev_b32_e32 v115, 1, v114\n\t" \
    "s_and_b64 vcc, vcc, %[sb]\n\t" \
    "v_add_u32_e32 v115, v115, %[c]\n\t" \
    "s_or_b64 %[sp], vcc, %[sp]\n\t" \
    "s_andn2_b64 %[sp], %[sp], %[r1m]\n\t"              /* role 0 withholds its next cursor when it is (all but) sure to need service */ \
    "v_cndmask_b32_e64 v115, v112, v115, %[sq]\n\t"     /* this lane's next cursor */ \
    "v_cndmask_b32_e64 v117, v115, -1, %[sp]\n\t" \
    "v_lshl_or_b32 v120, v113, 24, %[k8]\n\t" \
    "v_ffbh_u32_e32 v120, v120\n\t" \
    "v_mov_b32_dpp v122, v117 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   /* role 0's next cursor, in both lanes */ \
    "v_mov_b32_dpp v123, v117 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   /* role 1's */ \
    "v_cmp_eq_u32_e64 %[sc], v122, v119\n\t"            /* role 0's match ends where role 1 probed: both count */ \
    "v_lshrrev_b32_e32 v120, 1, v120\n\t" \
    "v_add_u32_e32 v118, 5, v96\n\t"                    /* role 1's cursor + 1 */ \
    SNK_SPEC_DEF_A \
    "v_cndmask_b32_e64 v122, v122, v123, %[sc]\n\t"     /* the chain's next cursor */ \
    "v_add_u32_e32 %[c], v122, %[five]\n\t"             /* this lane's */ \
    "v_sub_u32_e32 v116, %[c], %[rbc]\n\t" \
    "v_cmp_lt_u32_e64 %[sl], 15, v116\n\t" \
    "v_lshlrev_b32_e32 v116, 1, v116\n\t" \
    "v_cndmask_b32_e64 v118, v96, v118, %[sc]\n\t"      /* cursor + 1 of the last probe that counted */ \
    "s_waitcnt vmcnt(0)\n\t" \
    "v_cndmask_b32_e64 %[r0], %[r0], %[r1], %[sl]\n\t" \
    "v_cndmask_b32_e64 %[r1], %[r1], v108, %[sl]\n\t" \
    "v_alignbit_b32 %[wc], %[r1], %[r0], v116\n\t" \
    "v_lshrrev_b32_e32 v117, 7, %[wc]\n\t" \
    "v_and_b32_e32 v117, 0x7fe, v117\n\t" \
    "ds_read_u16 %[s1], v117\n\t" \
    "v_lshrrev_b32_e32 v121, 3, %[wc]\n\t" \
    "v_and_b32_e32 v121, 0x7fe, v121\n\t" \
    "ds_read_u16 %[ns2], v121\n\t" SNK_PADE \
    SNK_SPEC_DEF_B \
    SNK_SPEC_CURS \
    /* this probe's accounting; the masks of the next trip */ \
    /* the masks of the next trip and the exit test.  Service: the next cursor at the limit (incl. a straddling candidate and 12 equal */ \
    /* bases: a huge next cursor), or sp -- literal run, budget, back-extension of 4 -- which PRE / the compare above decided for   */ \
    /* role 0 (a superset of the exact test by the one case of a candidate within 4 bases of the stream start: a needless exit).  */ \
    /* The account of this probe (opn) is taken in the next trip's load shadow, or below when the loop ends here.                  */ \
    "v_cmp_ne_u32_e64 %[sm], v122, v118\n\t"            /* the last probe that counted was a match: put(cur-2) owed */ \
    "v_cmp_ge_u32_e64 %[st], v115, " LIM "\n\t" \
    "s_orn2_b64 %[ss], %[sc], %[r1m]\n\t"               /* role 0, and role 1 when its probe counts */ \
    "s_or_b64 %[sm], %[sm], %[r1m]\n\t" \
    "s_or_b64 vcc, %[st], %[sp]\n\t" \
    "s_and_b64 vcc, vcc, %[ss]\n\t" \
    "s_cbranch_vccz 1b\n\t" \
    SNK_SPEC_OPN \
    "s_waitcnt lgkmcnt(0)\n\t"
// (Round 4 pinned these VGPR operands for a while: the loop had drifted from 572 k to 560 k pair-compr./s as the code around
// it changed, and giving the operands the round-3 build's registers brought the rate back.  The cause was not the registers
// but WHERE THE LOOP LAY: with its head 24 bytes past a 32-byte boundary the same instructions run 2.3 % slower -- a sweep of
// the head's offset in steps of 8 bytes, DESIGN.md section 6.00 -- and either allocation runs at 572 k once the head is
// aligned, SNK_LOOP_ALIGN.  So the allocator keeps the choice.)
#define SNK_SPEC_OPERANDS SNK_SPEC_OPERANDS_FREE SNK_SPEC_CLOBBERS
#define SNK_SPEC_OPERANDS_X SNK_SPEC_OPERANDS_FREE SNK_SPEC_CLOBBERS
#define SNK_SPEC_OPERANDS_FREE \
    : [c] "+v"(c), [wc] "+v"(wc), [s1] "+v"(s1), [s2] "+v"(s2), [r0] "+v"(r0), [r1] "+v"(r1), \
      [rbc] "+v"(rbc), [nxoff] "+v"(nxoff), [anchor] "+v"(anchor_c), [op] "+v"(op), \
      [opn] "+v"(opn), [ns2] "+v"(ns2), [sm] "+s"(sm), [sl] "+s"(sl), [scm] "+s"(scm), [sc0] "+s"(sc0), [sq] "+s"(sq), [sc] "+s"(sc), \
      [t] "=&v"(t), [lit] "=&v"(lit), [sv] "=&s"(sv), [ss] "=&s"(ss), [st] "=&s"(st), [sp] "=&s"(sp), \
      [sb] "=&s"(sb), [ex] "=&s"(ex) \
    : [lb] "v"(lds_off), [five] "v"(five), [fivec] "v"(5u - five), [sx] "v"(sx), [kx] "v"(kx), \
      [xoffB] "v"(xoffB), [yoffB] "v"(yoffB), [T0] "v"(T0), [limc] "v"(limc), [oz] "v"(oz), [dm] "v"(DUMMY), \
      [k8] "s"(0x00800000u), [arena] "s"(arena), [marena] "s"(marena), [r1m] "s"(r1m), [vz] "v"(0u)
#define SNK_SPEC_CLOBBERS \
    : "memory", "vcc", "scc", "v88", "v89", \
      "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", \
      "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", \
      "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

// ---- the steady loop with speculative partner lanes ("fast_spec") ---------------------------------------------------
// A wave of the 2-bit kernel serves 21 chains with 64-lane instructions and is bound by its own instruction issue: 43
// lanes of every instruction do nothing.  They cannot run more chains (LDS), but they can run the SAME chain ahead.  On
// genome data two probes of three are followed by a probe exactly 5 bases on (the match of the 5-mer ends there), so here
// every chain has two lanes, a DPP pair (2i, 2i+1): lane 2i ("role 0") is the chain, lane 2i+1 ("role 1") probes at cur + 5
// in the same trip, as liblz4's immediate probe after a match would: owed put of cur + 3, no literals, no catch-up.  When
// role 0's probe turns out to be a match that ends exactly there (and neither needs service) both probes count and the
// chain's next cursor comes from role 1; otherwise role 1's work is dropped.  Exactness: role 1 READS the table with
// everybody at the top of the trip -- role 0's put(cur-2) has been issued before, its put(cur) is patched in by a select
// when the slots are equal -- and WRITES nothing until it is known to count: its two puts are issued at the top of the
// next trip, before anything else (liblz4's order).  Pure ACGT pairs only.  ASM: the hand-scheduled form above.
__device__ __forceinline__ uint32_t snk_pair_swap(uint32_t v)          // the partner's value inside the lane pair (2i, 2i+1)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
#define SNK_PAIR_TAKE(v) do { const uint32_t sw_ = snk_pair_swap((uint32_t)(v)); if (R1) (v) = sw_; } while (0)

template <bool ASM, bool EXC>
__device__ __forceinline__ void snk_fast_steady_spec_once(SnkFastLane &L, const bool R1, snk_g8 *const arena, snk_g8 *const marena,
                                                          uint16_t *tbl_, uint32_t *bm_, uint32_t lds_off, uint32_t round_bases SNK_PROF_ARG)
{
#ifdef SNK_STATS
    const unsigned long long stat_te = clock64();
#endif
#ifdef SNK_HOST_EMU
    // CPU emulation of the lane pair (tests/emu/): two host threads in lockstep; every LDS access of the loop is a
    // synchronisation point of the pair, so that the accesses keep the order the wave's instructions give them
    const uint16_t *const lut0 = (const uint16_t *)snk_lds8;
    SnkEmuLds<uint16_t> tbl(tbl_); SnkEmuLds<uint32_t> bm(bm_);
#else
    uint16_t *const tbl = tbl_; uint32_t *const bm = bm_;
    const SNK_AS3 uint16_t *const lut0 = (const SNK_AS3 uint16_t *)0;
#endif
    // role 1 takes the chain's state at the loop entry from role 0
    uint32_t vb = L.base, lx = L.s.lx, xoff = L.s.xoff, yoff = L.s.yoff, mfl1 = L.mfl1, olimit = L.olimit;
    uint32_t cur0 = L.cur, anchor0 = L.anchor, op = L.op, pend0 = L.pending ? 1u : 0u, wrb = L.w.rb, wsoff = L.w.soff, worg = L.w.org;
    // the chain's limit, as in the one-lane loop: block end, the reservoir's source, (EXC) the next exception site, a short round
    uint32_t lim_abs;
    {
        const uint32_t wlim = L.w.lim, limw = wlim == 0xFFFFFFFFu ? wlim : wlim + 1u;
        lim_abs = L.mfl1 < limw ? L.mfl1 : limw;
        if (EXC) lim_abs = lim_abs < L.xlim ? lim_abs : L.xlim;
        if (EXC && round_bases != 0xFFFFFFFFu && lim_abs - L.cur > round_bases) lim_abs = L.cur + round_bases;
    }
    // (EXC, wave-uniform) can a chain of this run read an entry whose window holds an exception?  Only then the class window is loaded.
    const bool need_mask = EXC && __any(!R1 && L.cur < L.mask_until);
    SNK_PAIR_TAKE(vb); SNK_PAIR_TAKE(lx); SNK_PAIR_TAKE(xoff); SNK_PAIR_TAKE(yoff); SNK_PAIR_TAKE(mfl1); SNK_PAIR_TAKE(lim_abs);
    SNK_PAIR_TAKE(olimit); SNK_PAIR_TAKE(cur0); SNK_PAIR_TAKE(anchor0); SNK_PAIR_TAKE(op); SNK_PAIR_TAKE(pend0);
    SNK_PAIR_TAKE(wrb); SNK_PAIR_TAKE(wsoff); SNK_PAIR_TAKE(worg);

    const int32_t T0 = (int32_t)(vb - 65536u);
    const int32_t X0 = T0 - 4, Y0 = T0 - 4 - (int32_t)lx;
    const uint32_t kx = (uint32_t)X0 & 3u;
    const uint32_t xoffB = xoff + (uint32_t)(X0 >> 2), yoffB = yoff + (uint32_t)(Y0 >> 2);
    const int32_t sx = (int32_t)lx - 11 - T0;
    const uint32_t limc = lim_abs - vb;
    const int32_t olimZ = (int32_t)olimit - (int32_t)SNK_FAST_ZONE + 10;
    const uint32_t DUMMY = SNK_FSLOTS - 1u;

    uint32_t c = cur0 - vb + (R1 ? 5u : 0u);                      // this lane's cursor
    uint32_t anchor_c = R1 ? c : anchor0 - vb;                    // role 1 never has literals
    // this lane's own reservoir, seated from the arena: c - rbc in [0, 16)
    uint32_t rbc = wrb + 4u - vb;
    uint32_t nxoff = wsoff + ((wrb + 32u - worg) >> 2);
    { const uint32_t sl = (c - rbc) >> 4; rbc += 16u * sl; nxoff += 4u * sl; }
    uint32_t r0 = snk_ld4g(arena + (size_t)(nxoff - 8u)), r1 = snk_ld4g(arena + (size_t)(nxoff - 4u)), r2 = 0u;
    uint32_t wc = __builtin_amdgcn_alignbit(r1, r0, 2u * (c - rbc));
    uint32_t s1 = lut0[(wc >> 8) & 1023u];
    uint32_t s2 = (R1 || pend0) ? (uint32_t)lut0[(wc >> 4) & 1023u] : DUMMY;
    uint32_t d1 = DUMMY, d2 = DUMMY, dc = 2u;                     // role 1: the puts of its probe of the last trip, if it counted
    uint32_t t; bool valid;

#ifdef SNK_STATS
    const unsigned long long stat_t0 = clock64();
    P.prologue += stat_t0 - stat_te;
#endif
#ifndef SNK_HOST_EMU
    if (ASM) {
        const uint64_t r1m = __builtin_amdgcn_ballot_w64(R1);                   // role 1 among the lanes in the loop
        const uint32_t five = R1 ? 5u : 0u;
        const uint32_t oz = (uint32_t)(olimZ - 14 - 3);
        uint32_t ns2 = lut0[(wc >> 4) & 1023u];
        uint32_t opn = op, lit;
        if (R1) op = 0u, opn = 0u;                                              // role 1 keeps no account
        uint64_t sm = __builtin_amdgcn_ballot_w64(R1 || pend0), sl = 0, scm = 0, sc0 = 0, sq = 0, sc = 0, sv, ss, st, sp, sb, ex;   // (sq, sc: nothing to commit in the first trip)
        if (!EXC) {
            if (__all(sx + 15 <= 0))
                asm volatile(SNK_SPEC_TABLE SNK_STEADY_ADDR_YONLY SNK_SPEC_SHADOW("%[t]") SNK_SPEC_PRE("") SNK_SPEC_REST("%[limc]")
                             SNK_SPEC_OPERANDS);
            else
                asm volatile(SNK_SPEC_TABLE SNK_STEADY_ADDR_DUAL SNK_SPEC_SHADOW("v103") SNK_STEADY_STRADDLE
                             SNK_SPEC_PRE("s_or_b64 %[sp], %[sp], %[ss]\n\t") SNK_SPEC_REST("v105") SNK_SPEC_OPERANDS);
        } else if (!need_mask) {
            if (__all(sx + 15 <= 0))
                asm volatile(SNK_SPEC_TABLE SNK_STEADY_ADDR_YONLY SNK_SPEC_SHADOW("%[t]") SNK_SPEC_PRE("") SNK_SPEC_REST("%[limc]")
                             SNK_SPEC_OPERANDS_X);
            else
                asm volatile(SNK_SPEC_TABLE SNK_STEADY_ADDR_DUAL SNK_SPEC_SHADOW("v103") SNK_STEADY_STRADDLE
                             SNK_SPEC_PRE("s_or_b64 %[sp], %[sp], %[ss]\n\t") SNK_SPEC_REST("v105") SNK_SPEC_OPERANDS_X);
        } else {
            if (__all(sx + 15 <= 0))
                asm volatile(SNK_SPEC_TABLE SNK_STEADY_ADDR_YONLY SNK_SPEC_SHADOW_X("%[t]", SNK_SPEC_MASKLOAD) SNK_SPEC_PRE("")
                             SNK_SPEC_REST_X("%[limc]", SNK_SPEC_MASKOR, "1") SNK_SPEC_OPERANDS_X);
            else
                asm volatile(SNK_SPEC_TABLE SNK_STEADY_ADDR_DUAL SNK_SPEC_SHADOW_X("v103", SNK_SPEC_MASKLOAD) SNK_STEADY_STRADDLE
                             SNK_SPEC_PRE("s_or_b64 %[sp], %[sp], %[ss]\n\t") SNK_SPEC_REST_X("v105", SNK_SPEC_MASKOR, "1") SNK_SPEC_OPERANDS_X);
        }
#ifdef SNK_STATS
        const unsigned long long stat_t1 = clock64();
        P.loop += stat_t1 - stat_t0; P.entries++;
#endif
        // the loop leaves before the trip's probes are committed: c holds the next cursor, lit + anchor the current one
        uint32_t ccur = R1 ? anchor_c : lit + anchor_c;
        const uint32_t t1 = snk_pair_swap(t), c1 = snk_pair_swap(ccur);
        if (R1) return;
        if ((sc >> (threadIdx.x & 63u)) & 1ull) {      // role 1's probe counts: role 0's is an ordinary 5-base match; hand over role 1's
            op = opn; anchor_c = c1; ccur = c1; t = t1;
        }
        L.cur = vb + ccur; L.anchor = vb + anchor_c; L.op = op; L.step = 1u; L.nb = 63u + (ccur - anchor_c);
        L.w.rb = 0x80000000u;
        if (EXC) snk_fast_exc_handover(L, vb + ccur, (uint32_t)(T0 + (int32_t)t), t > ccur);
        else     snk_fast_finish(L, vb + ccur, (uint32_t)(T0 + (int32_t)t), t > ccur);
#ifdef SNK_STATS
        P.finish += clock64() - stat_t1;
#endif
        return;
    }
#endif
    for (;;) {
        // ---- table, in liblz4's order for the chain: role 1's probe of the last trip (put, put), then role 0's (put, get, put);
        //      role 1 only reads at its new cursor.  A lane that has nothing to write writes the unused slot.
        tbl[d2] = (uint16_t)(dc - 2u); atomicOr(&bm[d2 >> 5], 1u << (d2 & 31u));
        tbl[d1] = (uint16_t)dc;        atomicOr(&bm[d1 >> 5], 1u << (d1 & 31u));
        const uint32_t b2 = R1 ? DUMMY : s2, b1 = R1 ? DUMMY : s1;
        tbl[b2] = (uint16_t)(c - 2u);  atomicOr(&bm[b2 >> 5], 1u << (b2 & 31u));
        const uint32_t bit1 = s1 & 31u;
        const uint32_t e = tbl[s1];
        const uint32_t bw = atomicOr(&bm[s1 >> 5], (R1 ? 0u : 1u) << bit1);
        tbl[b1] = (uint16_t)c;
        t = e + (((bw >> bit1) & 1u) << 16);
        {   // role 1: role 0's put(cur) of this trip and its own owed put(cur-2) have not been written
            const uint32_t ps1 = snk_pair_swap(s1);
            t = (R1 && ps1 == s1) ? 65536u + c - 5u : t;
            t = (R1 && s2 == s1) ? 65534u + c : t;
        }
        valid = t > c;

        // ---- candidate window and the reservoir refill ----
        const bool inx = (int32_t)t < sx;
        const uint32_t tt = t + (inx ? kx : 0u);
        const uint64_t v = snk_ld8g(arena + (size_t)((inx ? xoffB : yoffB) + (tt >> 2)));
        r2 = snk_ld4g(arena + (size_t)nxoff);
        const bool straddle = valid & ((uint32_t)((int32_t)t - sx) < 15u);
        const uint32_t lit = c - anchor_c;
        const uint32_t wd = __builtin_amdgcn_alignbit((uint32_t)(v >> 32), (uint32_t)v, 2u * (tt & 3u));
        uint32_t wm = 0u;                                         // (EXC) the candidate's class window: 11 where a byte is not one of ACGT
        if (EXC && need_mask) {
            const uint64_t mv = snk_ld8g(marena + (size_t)((inx ? xoffB : yoffB) + (tt >> 2)));
            wm = __builtin_amdgcn_alignbit((uint32_t)(mv >> 32), (uint32_t)mv, 2u * (tt & 3u));
        }

        // ---- compare, this lane's next cursor ----
        const uint32_t x = (wc ^ wd) | wm;
        const uint32_t r = snk_ffbl(x >> 8);
        const bool m = valid & (r >= 8u);
        const uint32_t e2 = c + (r >> 1);
        const uint32_t ncur = m ? e2 : c + 1u;

        // ---- this probe's accounting ----
        const uint32_t eq = (uint32_t)__builtin_clz(((x & 0xFFu) << 24) | 0x00800000u) >> 1;
        uint32_t b = eq < lit ? eq : lit;
        { const uint32_t cand = (uint32_t)(T0 + (int32_t)t); b = b < cand ? b : cand; }
        const uint32_t opn = op + (lit - b) + 3u;
        int32_t mx = (int32_t)(b + 11u) > (int32_t)lit ? (int32_t)(b + 11u) : (int32_t)lit;
        { const int32_t z = (int32_t)op - olimZ + 14 + 3; mx = mx > z ? mx : z; }       // (+3: role 1's sequence may be added in this trip)
        const bool svc = (mx >= 15) | (ncur >= limc) | straddle | (e2 >= 0x20000000u);

        // ---- the pair decides: role 0 offers "matched, ends where role 1 probed, no service", role 1 "no service" ----
        const bool offer = !svc & (R1 | (m & ((r >> 1) == 5u)));
        const uint32_t pk = (ncur & 0x3FFFFFFFu) | (m ? 0x40000000u : 0u) | (offer ? 0x80000000u : 0u);
        const uint32_t qk = snk_pair_swap(pk);
        const bool com = offer & ((int32_t)qk < 0);               // role 1's probe counts (the same value in both lanes)
        const uint32_t n0 = R1 ? qk : pk, n1 = R1 ? pk : qk;      // (ncur, m) of role 0 / of role 1
        const uint32_t nn = com ? n1 : n0;
        const uint32_t cp = nn & 0x3FFFFFFFu;                     // the chain's next cursor
        const bool mp = (nn & 0x40000000u) != 0u;                 // ... after a match (put(cur-2) owed)

#ifdef SNK_STATS
        {   // wave trips, why chains ask for service (as the one-lane loop counts them), and how often the second lane counts
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
            if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(act)) SNK_COUNT(15);
            if (!R1) SNK_COUNT(56);
            if (!R1 && com) SNK_COUNT(57);
            if (!R1 && svc) {
                if ((int32_t)lit >= 15) SNK_COUNT(16);
                if (b >= 4u) SNK_COUNT(17);
                if ((int32_t)op - olimZ + 14 + 3 >= 15) SNK_COUNT(18);
                if (r == 0xFFFFFFFFu && valid) SNK_COUNT(19);
                else if (ncur >= limc) { if (vb + ncur >= mfl1) SNK_COUNT(20); else SNK_COUNT(21); }
                if (straddle) SNK_COUNT(22);
            }
        }
#endif
        // one wave-uniform exit, decided by the chains (role 0)
        if (__builtin_amdgcn_ballot_w64(svc & !R1) != 0ull) break;

        // ---- commit ----
        const bool m1 = (n1 & 0x40000000u) != 0u;
        op = m ? opn : op; anchor_c = m ? ncur : anchor_c;                       // role 0's probe (role 1's own copies are not used)
        if (com & m1) { op += 3u; anchor_c = n1 & 0x3FFFFFFFu; }                 // role 1's: token + offset, no literals
        d1 = (R1 & com) ? s1 : DUMMY; d2 = (R1 & com) ? s2 : DUMMY; dc = c;
        const uint32_t nc = cp + (R1 ? 5u : 0u);
        const uint32_t no = nc - rbc;                                            // 0..31
        const bool sl = no >= 16u;
        const uint32_t lo = sl ? r1 : r0, hi = sl ? r2 : r1;
        const uint32_t nwc = __builtin_amdgcn_alignbit(hi, lo, 2u * no);
        const uint32_t ns1 = lut0[(nwc >> 8) & 1023u];
        const uint32_t ns2 = lut0[(nwc >> 4) & 1023u];
        r0 = lo; r1 = hi; rbc += sl ? 16u : 0u; nxoff += sl ? 4u : 0u;
        c = nc; wc = nwc; s1 = ns1; s2 = (R1 | mp) ? ns2 : DUMMY;
        anchor_c = R1 ? nc : anchor_c;
    }
    if (R1) return;
    // role 0: hand the chain over in the state "table operations of the probe at c done, match not evaluated"
    L.cur = vb + c; L.anchor = vb + anchor_c; L.op = op; L.step = 1u; L.nb = 63u + (c - anchor_c);
    L.w.rb = 0x80000000u;
    if (EXC) snk_fast_exc_handover(L, vb + c, (uint32_t)(T0 + (int32_t)t), valid);
    else     snk_fast_finish(L, vb + c, (uint32_t)(T0 + (int32_t)t), valid);
}

// A chain that leaves the loop inside a block seats its reservoir again from the source it was reading.
__device__ __forceinline__ void snk_fast_reseat_same(SnkFastLane &L)
{
    const uint32_t cur = L.cur, lx = L.s.lx;
    if (cur + L.step > L.mfl1) return;
    if (cur >= lx + 4u && L.w.org == lx && L.w.soff == L.s.yoff && L.w.lim == 0xFFFFFFFFu)
        snk_win_init(L.w, L.s.arena, L.s.yoff, lx, 0xFFFFFFFFu, cur);
    else if (cur >= 4u && cur + 12u <= lx && L.w.org == 0u && L.w.soff == L.s.xoff && L.w.lim == lx - 12u)
        snk_win_init(L.w, L.s.arena, L.s.xoff, 0u, lx - 12u, cur);
}

// The two-lane loop with a quick turn-around (round 4, pure ACGT): when every chain of the run can go straight back into the loop
// after its hand-over -- inside its block, the reservoir's source unchanged, nothing for the general path to do: a match that ran
// past the 12-base window, a long back-extension, a reservoir at its source's limit... -- the wave does, from here, instead of
// through the head of the wave loop, whose compiled generality (job hand-out, the rounds and their votes, the block step: ~1 300
// vector instructions per pass at one wave per SIMD) was 4 k of such an exit's 6 k cycles.  A block end, the seam, a finished
// frame, a lane that needs a general probe: back to the head as before.
// Block ends too (half of the exits on genome data): a chain whose block is followed by another full-sized step of the same frame
// -- no snapshot to write, not the frame's end -- takes the block step right here; it seats the new block's reservoir itself.
template <bool ASM, bool EXC>
__device__ __forceinline__ void snk_fast_steady_spec(SnkFastLane &L, const bool R1, const SnkTables &T, const uint16_t *slot, uint32_t *out, uint32_t *status,
                                                     uint16_t *tbl_, uint32_t *bm_, uint32_t lds_off, uint32_t round_bases SNK_PROF_ARG)
{
    snk_g8 *const arena = (snk_g8 *)T.packed_arena, *const marena = (snk_g8 *)(EXC ? T.mask_arena : T.packed_arena);
    for (;;) {
        snk_fast_steady_spec_once<ASM, EXC>(L, R1, arena, marena, tbl_, bm_, lds_off, round_bases SNK_PROF_PASS);
        if (EXC) return;
        bool again = true;
        if (!R1) {
            if (L.cur + L.step > L.mfl1 && L.in_block && L.iend < L.n && L.n - L.iend >= 13u && L.blocks_left != 0u &&
                !(L.snap != 0 && L.iend == L.spos))
                (void)snk_fast_block_step<false, false>(L, T, tbl_, bm_, nullptr, slot, out, status);      // (returns false: a block follows)
            if (!snk_fast_eligible<false>(L)) snk_fast_reseat_same(L);
            again = snk_fast_eligible<false>(L);
        }
        if (!__all(again)) return;                 // (the lanes of this run; role 1 goes where its chain goes)
    }
}

#ifndef SNK_HOST_EMU
// ---- three lanes per chain (round 4; C++ statement, option fast_spec = 3) -------------------------------------------------
// Rows of 16 lanes hold 5 chains x 3 lanes (lane 15 of every row idles: 20 chains per wave): role 0 is the chain, role 1
// probes 5 bases ahead as in snk_fast_steady_spec, role 2 TEN bases ahead -- where the chain stands when two 5-base matches
// follow each other (42 % of the trips on genome data: 2.08 probes per chain-trip instead of 1.65, tools/next_probe_stats.py).
// Role 2 reads the table with everybody; the four puts the chain would have made between the read and its probe -- role 0's
// put(cur), role 1's put(cur+3) and put(cur+5), its own owed put(cur+8) -- are patched into what it read, latest first.  It
// counts when role 1 counts AND role 1's probe is a 5-base match that needs no service; its two puts are then made at the top
// of the next trip, after role 1's (liblz4's order).  Exactness never rests on a guess: a probe that does not count is dropped.
// Values travel inside the row by DPP row shifts (row_shr:n -- lane l reads lane l-n of its row; row_shl:n -- lane l+n).
__device__ __forceinline__ uint32_t snk_row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t snk_row_shr2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x112, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t snk_row_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x101, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t snk_row_shl2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x102, 0xF, 0xF, true); }
#define SNK_TRI_TAKE(v) do { const uint32_t a1_ = snk_row_shr1((uint32_t)(v)), a2_ = snk_row_shr2((uint32_t)(v)); \
                             (v) = R == 1u ? a1_ : R == 2u ? a2_ : (v); } while (0)

// P2 = 10: role 2 as described.  P2 = 6: the other placement -- role 2 probes SIX bases ahead and counts when role 0's match ends
// there (17 % of the trips; role 1's then does not): its probe follows role 0's directly, so it is patched like role 1's
// (role 0's put(cur), its own owed put) and at most one of the two probes ahead counts in a trip.
template <uint32_t P2>
__device__ __forceinline__ void snk_fast_steady_spec3(SnkFastLane &L, const uint32_t R, snk_g8 *const arena,
                                                      uint16_t *tbl, uint32_t *bm SNK_PROF_ARG)
{
    static_assert(P2 == 10u || P2 == 6u, "role 2 probes 10 or 6 bases ahead");
    const uint32_t ahead = R == 1u ? 5u : R == 2u ? P2 : 0u;          // this lane's cursor = the chain's + ahead
#ifdef SNK_STATS
    const unsigned long long stat_te = clock64();
#endif
    const SNK_AS3 uint16_t *const lut0 = (const SNK_AS3 uint16_t *)0;
    // roles 1 and 2 take the chain's state at the loop entry from role 0
    uint32_t vb = L.base, lx = L.s.lx, xoff = L.s.xoff, yoff = L.s.yoff, olimit = L.olimit;
    uint32_t cur0 = L.cur, anchor0 = L.anchor, op = L.op, pend0 = L.pending ? 1u : 0u, wrb = L.w.rb, wsoff = L.w.soff, worg = L.w.org;
    uint32_t lim_abs;
    {
        const uint32_t wlim = L.w.lim, limw = wlim == 0xFFFFFFFFu ? wlim : wlim + 1u;
        lim_abs = L.mfl1 < limw ? L.mfl1 : limw;
    }
    SNK_TRI_TAKE(vb); SNK_TRI_TAKE(lx); SNK_TRI_TAKE(xoff); SNK_TRI_TAKE(yoff); SNK_TRI_TAKE(lim_abs);
    SNK_TRI_TAKE(olimit); SNK_TRI_TAKE(cur0); SNK_TRI_TAKE(anchor0); SNK_TRI_TAKE(op); SNK_TRI_TAKE(pend0);
    SNK_TRI_TAKE(wrb); SNK_TRI_TAKE(wsoff); SNK_TRI_TAKE(worg);

    const int32_t T0 = (int32_t)(vb - 65536u);
    const int32_t X0 = T0 - 4, Y0 = T0 - 4 - (int32_t)lx;
    const uint32_t kx = (uint32_t)X0 & 3u;
    const uint32_t xoffB = xoff + (uint32_t)(X0 >> 2), yoffB = yoff + (uint32_t)(Y0 >> 2);
    const int32_t sx = (int32_t)lx - 11 - T0;
    const uint32_t limc = lim_abs - vb;
    const int32_t olimZ = (int32_t)olimit - (int32_t)SNK_FAST_ZONE + 10;
    const uint32_t DUMMY = SNK_FSLOTS - 1u;
    const uint32_t CM = 0x1FFFFFFFu;                               // cursor bits of the packed (next cursor, flags) word

    uint32_t c = cur0 - vb + ahead;                                // this lane's cursor
    uint32_t anchor_c = R ? c : anchor0 - vb;                      // roles 1 and 2 never have literals
    uint32_t rbc = wrb + 4u - vb;
    uint32_t nxoff = wsoff + ((wrb + 32u - worg) >> 2);
    { const uint32_t sl = (c - rbc) >> 4; rbc += 16u * sl; nxoff += 4u * sl; }
    uint32_t r0 = snk_ld4g(arena + (size_t)(nxoff - 8u)), r1 = snk_ld4g(arena + (size_t)(nxoff - 4u)), r2 = 0u;
    uint32_t wc = __builtin_amdgcn_alignbit(r1, r0, 2u * (c - rbc));
    uint32_t s1 = lut0[(wc >> 8) & 1023u];
    uint32_t s2 = (R || pend0) ? (uint32_t)lut0[(wc >> 4) & 1023u] : DUMMY;
    uint32_t d1 = DUMMY, d2 = DUMMY, dc = 2u;                     // roles 1 / 2: the puts of their probe of the last trip, if it counted
    uint32_t t; bool valid;
#ifdef SNK_STATS
    const unsigned long long stat_t0 = clock64();
    P.prologue += stat_t0 - stat_te;
#endif
    for (;;) {
        // ---- the chain's table operations in liblz4's order: role 1's probe of the last trip (put, put), role 2's (put, put),
        //      role 0's (put, get, put); roles 1 and 2 only read at their new cursors.  Nothing to write: the unused slot.
        { const uint32_t x2 = R == 1u ? d2 : DUMMY, x1 = R == 1u ? d1 : DUMMY;
          tbl[x2] = (uint16_t)(dc - 2u); atomicOr(&bm[x2 >> 5], 1u << (x2 & 31u));
          tbl[x1] = (uint16_t)dc;        atomicOr(&bm[x1 >> 5], 1u << (x1 & 31u)); }
        { const uint32_t x2 = R == 2u ? d2 : DUMMY, x1 = R == 2u ? d1 : DUMMY;
          tbl[x2] = (uint16_t)(dc - 2u); atomicOr(&bm[x2 >> 5], 1u << (x2 & 31u));
          tbl[x1] = (uint16_t)dc;        atomicOr(&bm[x1 >> 5], 1u << (x1 & 31u)); }
        const uint32_t b2 = R ? DUMMY : s2, b1 = R ? DUMMY : s1;
        tbl[b2] = (uint16_t)(c - 2u);  atomicOr(&bm[b2 >> 5], 1u << (b2 & 31u));
        const uint32_t bit1 = s1 & 31u;
        const uint32_t e = tbl[s1];
        const uint32_t bw = atomicOr(&bm[s1 >> 5], (R ? 0u : 1u) << bit1);
        tbl[b1] = (uint16_t)c;
        t = e + (((bw >> bit1) & 1u) << 16);
        {   // what the lanes before this one will have put by the time the chain reaches this lane's cursor (latest last)
            const uint32_t a1s1 = snk_row_shr1(s1), a1s2 = snk_row_shr1(s2), a2s1 = snk_row_shr2(s1);
            if (P2 == 10u) {
                t = (R == 2u && a2s1 == s1) ? 65536u + c - 10u : t;      // role 0's put(cur)
                t = (R == 2u && a1s2 == s1) ? 65536u + c - 7u : t;       // role 1's owed put(cur+3)
                t = (R != 0u && a1s1 == s1) ? 65536u + c - 5u : t;       // the put(cursor) of the lane before
            } else {
                t = (R == 1u && a1s1 == s1) ? 65536u + c - 5u : t;       // role 0's put(cur), for either lane ahead
                t = (R == 2u && a2s1 == s1) ? 65536u + c - 6u : t;
            }
            t = (R != 0u && s2 == s1) ? 65534u + c : t;                   // the own owed put
        }
        valid = t > c;

        // ---- candidate window and the reservoir refill ----
        const bool inx = (int32_t)t < sx;
        const uint32_t tt = t + (inx ? kx : 0u);
        const uint64_t v = snk_ld8g(arena + (size_t)((inx ? xoffB : yoffB) + (tt >> 2)));
        r2 = snk_ld4g(arena + (size_t)nxoff);
        const bool straddle = valid & ((uint32_t)((int32_t)t - sx) < 15u);
        const uint32_t lit = c - anchor_c;
        const uint32_t wd = __builtin_amdgcn_alignbit((uint32_t)(v >> 32), (uint32_t)v, 2u * (tt & 3u));

        // ---- compare, this lane's next cursor, its account ----
        const uint32_t x = wc ^ wd;
        const uint32_t r = snk_ffbl(x >> 8);
        const bool m = valid & (r >= 8u);
        const uint32_t e2 = c + (r >> 1);
        const uint32_t ncur = m ? e2 : c + 1u;
        const uint32_t eq = (uint32_t)__builtin_clz(((x & 0xFFu) << 24) | 0x00800000u) >> 1;
        uint32_t b = eq < lit ? eq : lit;
        { const uint32_t cand = (uint32_t)(T0 + (int32_t)t); b = b < cand ? b : cand; }
        const uint32_t opn = op + (lit - b) + 3u;
        int32_t mx = (int32_t)(b + 11u) > (int32_t)lit ? (int32_t)(b + 11u) : (int32_t)lit;
        if (R == 0u) { const int32_t z = (int32_t)op - olimZ + 14 + 6; mx = mx > z ? mx : z; }     // (+6: two more sequences may be added in this trip)
        const bool svc = (mx >= 15) | (ncur >= limc) | straddle | (e2 >= 0x10000000u);

        // ---- the lanes of the chain decide: every lane offers "no service"; a lane's match "ends at +5" ----
        // (bit 29: the match ends where the next lane probed -- for role 0 of the P2 = 6 form bit 28 says "6 bases on")
        const bool end5 = m & ((r >> 1) == 5u), end6 = m & ((r >> 1) == 6u);
        const uint32_t CMX = P2 == 6u ? 0x0FFFFFFFu : CM;
        const uint32_t pk = (ncur & CMX) | ((P2 == 6u && end6) ? 0x10000000u : 0u) | (end5 ? 0x20000000u : 0u) | (m ? 0x40000000u : 0u) | (svc ? 0u : 0x80000000u);
        const uint32_t A1 = snk_row_shr1(pk), A2 = snk_row_shr2(pk), B1 = snk_row_shl1(pk), B2 = snk_row_shl2(pk);
        const uint32_t q0 = R == 0u ? pk : R == 1u ? A1 : A2;      // (next cursor, flags) of role 0 / 1 / 2, the same in all three lanes
        const uint32_t q1 = R == 0u ? B1 : R == 1u ? pk : A1;
        const uint32_t q2 = R == 0u ? B2 : R == 1u ? B1 : pk;
        const bool com1 = ((q0 & 0xA0000000u) == 0xA0000000u) & ((int32_t)q1 < 0);           // role 1's probe counts
        // role 2's counts when role 1's is a 5-base match too -- and the chain's advance stays within what every lane's
        // 32-base reservoir can follow in one slide (16 bases: a role-2 match of up to 6)
        const uint32_t c0 = c - ahead;
        const bool com2 = P2 == 10u ? (com1 & ((q1 & 0x20000000u) != 0u) & ((int32_t)q2 < 0) & ((q2 & CM) - c0 <= 16u))
                                    : (((q0 & 0x90000000u) == 0x90000000u) & ((int32_t)q2 < 0) & ((q2 & CMX) - c0 <= 16u));
        const uint32_t nn = com2 ? q2 : com1 ? q1 : q0;
        const uint32_t cp = nn & CMX;                             // the chain's next cursor
        const bool mp = (nn & 0x40000000u) != 0u;                 // ... after a match (put(cur-2) owed)
#ifdef SNK_STATS
        {
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
            if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(act)) SNK_COUNT(15);
            if (R == 0u) SNK_COUNT(56);
            if (R == 0u && com1) SNK_COUNT(57);
            if (R == 0u && com2) SNK_COUNT(58);
        }
#endif
        // one wave-uniform exit, decided by the chains (role 0)
        if (__builtin_amdgcn_ballot_w64(svc & (R == 0u)) != 0ull) break;

        // ---- commit ----
        op = m ? opn : op; anchor_c = m ? ncur : anchor_c;                       // role 0's probe (the other roles' copies are not used)
        if (com1 & ((q1 & 0x40000000u) != 0u)) { op += 3u; anchor_c = q1 & CMX; } // role 1's: token + offset, no literals
        if (com2 & ((q2 & 0x40000000u) != 0u)) { op += 3u; anchor_c = q2 & CMX; } // role 2's
        const bool mine = ((R == 1u) & com1) | ((R == 2u) & com2);
        d1 = mine ? s1 : DUMMY; d2 = mine ? s2 : DUMMY; dc = c;
        const uint32_t nc = cp + ahead;
        const uint32_t no = nc - rbc;                                            // 0..31
        const bool sl = no >= 16u;
        const uint32_t lo = sl ? r1 : r0, hi = sl ? r2 : r1;
        const uint32_t nwc = __builtin_amdgcn_alignbit(hi, lo, 2u * no);
        const uint32_t ns1 = lut0[(nwc >> 8) & 1023u];
        const uint32_t ns2 = lut0[(nwc >> 4) & 1023u];
        r0 = lo; r1 = hi; rbc += sl ? 16u : 0u; nxoff += sl ? 4u : 0u;
        c = nc; wc = nwc; s1 = ns1; s2 = (R | (mp ? 1u : 0u)) ? ns2 : DUMMY;
        anchor_c = R ? nc : anchor_c;
    }
#ifdef SNK_STATS
    const unsigned long long stat_t1 = clock64();
    P.loop += stat_t1 - stat_t0; P.entries++;
#endif
    if (R) return;
    // role 0: hand the chain over in the state "table operations of the probe at c done, match not evaluated"
    L.cur = vb + c; L.anchor = vb + anchor_c; L.op = op; L.step = 1u; L.nb = 63u + (c - anchor_c);
    L.w.rb = 0x80000000u;
    snk_fast_finish(L, vb + c, (uint32_t)(T0 + (int32_t)t), valid);
#ifdef SNK_STATS
    P.finish += clock64() - stat_t1;
#endif
}
#endif

// How job numbers map to ordered pairs, and how the waves of a launch share them.
//   jobs != NULL : explicit list (mixed tiles, pair lists, the single-sequence pass).
//   jobs == NULL : dense tile of the N x N matrix: job q is the pair (x = r0 + q % rows, y = q / rows),
//                  its size goes to out[(x - r0) * n + y].  Suffix-major, so consecutive jobs share y.
// Work is handed out per WAVE in batches of `batch` consecutive jobs (= the lanes of a wave).  `queue` NULL (uniform
// lengths): wave w of the launch starts with batch w, its later batches are w + k * (waves of the launch) -- the waves
// of a workgroup keep walking the same suffix y, the L1 stays hot.  `queue` set (ragged lengths, far chains): *queue
// counts the JOBS handed out (starts at 0); a wave adds the number of its chains and takes that many consecutive jobs,
// so no wave idles while jobs are left.  Inside a wave a lane that finishes its pair takes the next job of the wave's
// batch at once.
struct SnkFastGrid {
    const SnkJob *jobs;
    uint32_t n_jobs;
    uint32_t r0, rows, n;
    uint32_t batch;
    uint32_t *queue;          // jobs handed out so far (starts at 0); NULL = static round robin
    const uint32_t *yorder;   // dense tile: column visited k-th (longest suffix first when lengths are ragged: the
                              // big jobs go out first and the launch ends on small ones); NULL = column k
    // ---- chains beyond the LDS (see snk_fast_kernel_body): waves [lds_waves, blockDim.x / 64) of a workgroup keep their
    // chains' tables in global memory.  far_lanes = 0: none (every wave is an LDS wave).
    uint32_t *far_tab = nullptr;   // [workgroups][far waves][far_lanes][SNK_FSLOTS] u32 absolute positions
    uint32_t lds_waves = 0, far_lanes = 0;
    uint32_t far_stop = 0;    // a far wave takes no further jobs once fewer than this many are left in the queue (the
                              // LDS waves finish them sooner than a far chain would)
    uint32_t short_last = 0;  // 1: the last wave of every workgroup runs one chain fewer (sets with other-case stretches: the
                              // second LUT leaves room for 83 chains, not 84 -- 21 + 21 + 21 + 20)
    uint32_t flut = SNK_FLUT_B;   // LDS bytes in front of the chains: the slot LUT, and (sets with other-case letters) the other case's
};

__device__ __forceinline__ SnkJob snk_fast_job(const SnkFastGrid &G, uint32_t q)
{
    if (G.jobs) return G.jobs[q];
    SnkJob jb;
    const uint32_t yk = q / G.rows, xr = q - yk * G.rows;
    const uint32_t yi = G.yorder ? G.yorder[yk] : yk;
    jb.xi = (int32_t)(G.r0 + xr); jb.yi = (int32_t)yi; jb.out_idx = xr * G.n + yi; jb.snap = 0;
    return jb;
}

// Lane state at the start of a job (the table in LDS is initialised by snk_fast_kernel_body).
__device__ __forceinline__ void snk_fast_lane_init(SnkFastLane &L, const SnkTables &T, const SnkJob job)
{
    const uint32_t lx = T.len[job.xi];
    const uint32_t ly = job.yi >= 0 ? T.len[job.yi] : 0u;
    L.s.arena = (snk_g8 *)T.packed_arena;
    L.s.marena = (snk_g8 *)T.mask_arena;
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
    L.k3 = (0u - lx) & 3u;
    L.mfl1 = 0; L.mlimit = 0; L.olimit = 0; L.base = L.pos - L.k3; L.endcode = 0;     // mfl1 = 0: first iteration opens a block
    L.pending = false;
    L.w.soff = L.s.xoff; L.w.org = 0u; L.w.rb = 0x80000000u; L.w.lim = 0u; L.w.r0 = L.w.r1 = L.w.nx = 0u;
}

// Persistent workgroup: `waves` waves of `lanes` chains; dynamic LDS = 2 KiB LUT + 1904 B per chain.
// Every wave runs ONE flat loop: hand jobs to lanes that have none, general probes until every
// working lane is eligible, steady loop until some lane needs service.  The wave leaves when its
// lanes are idle and no job is left.
//
// Chains beyond the LDS (FAR).  The LDS holds 84 chains per CU and a lone wave per SIMD leaves the SIMD's issue slots
// half empty (a wave issues a VALU instruction every ~4 cycles, the SIMD could take one every 2), so a workgroup may
// carry extra waves whose chains keep their tables in GLOBAL memory, sized to stay resident in the XCD's L2
// (SnkFastGrid::far_*): u32 absolute positions, 3584 B per chain, one dependent load and two fire-and-forget stores per
// probe.  Same probe semantics, same job stream (dynamic queue), same code around the loop; the LDS waves are untouched.
// Not for sequences with exceptions, not for the singles pass (EXC / snapshot dumps know the LDS layout only).
// SPEC: chain i of the wave = the lane pair (2i, 2i+1), see snk_fast_steady_spec; TRI: three lanes of a 16-lane row, see snk_fast_steady_spec3
template <bool ASM, bool EXC, bool FAR, bool SPEC = false, uint32_t TRI = 0u>      // TRI: 0, or how far ahead role 2 probes (10 / 6)
__device__ __forceinline__ void snk_fast_wave(const SnkTables &T, const SnkFastGrid &G,
                                              uint32_t lanes, uint32_t *out, uint32_t *status)
{
    static_assert(!SPEC || !FAR, "two lanes per chain: the chains with their tables in LDS");
    static_assert(!TRI || (!SPEC && !FAR && !EXC && !ASM), "three lanes per chain: the C++ statement, pure ACGT");
#ifndef SNK_HOST_EMU
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
#endif
    uint16_t *slot = (uint16_t *)snk_lds8;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t waves = blockDim.x >> 6;

#ifdef SNK_STATS
    const unsigned long long stat_w0 = clock64(), stat_wall0 = wall_clock64();      // (wall clock: constant 100 MHz -> the shader clock of the run)
    SnkProf P = {};    // (lanes outside the loop's EXEC mask miss its share: lane 0 of a wave is
                                                                 //  nearly always inside -- good enough for an account)
#endif
    // the chain of the wave this lane belongs to (TRI: rows of 16 lanes = 5 chains x 3 lanes, lane 15 idles), and its role in it
    const uint32_t role = TRI ? ((lane & 15u) == 15u ? 3u : (lane & 15u) % 3u) : SPEC ? (lane & 1u) : 0u;
    const uint32_t cidx = TRI ? (lane >> 4) * 5u + (lane & 15u) / 3u : SPEC ? lane >> 1 : lane;
    const bool lane_on = role == 0u && cidx < lanes - ((G.short_last && wave + 1u == waves) ? 1u : 0u) &&
                         (!TRI || (lane & 15u) != 15u);                            // ... and runs (the other roles only inside the steady loop)
    // LDS waves: the chain's table at LDS address mine_off (dynamic LDS starts at 0).  FAR waves: at word far_idx * 896 of G.far_tab.
    // (when the resident set has letters of the other case the kernels keep TWO LUTs in front of the chains: G.flut = 4096)
    const uint32_t flut = EXC ? G.flut : SNK_FLUT_B;
    const uint32_t mine_off = FAR ? 0u : flut + (wave * lanes + (cidx < lanes ? cidx : 0u)) * SNK_FCHAIN_B;
    uint8_t *const mine = snk_lds8 + mine_off;
    uint16_t *const tbl = (uint16_t *)mine;
    uint32_t *const bm = (uint32_t *)(mine + SNK_FSLOTS * 2u);
    const uint32_t far_wave0 = FAR ? (blockIdx.x * (waves - G.lds_waves) + (wave - G.lds_waves)) * lanes : 0u;    // first chain of this wave
    uint32_t *const gt = FAR ? G.far_tab + (size_t)(far_wave0 + (lane_on ? lane : 0u)) * SNK_FSLOTS : nullptr;
    const uint32_t gtb = FAR ? (far_wave0 + (lane_on ? lane : 0u)) * (SNK_FSLOTS * 4u) : 0u;                   // ... as a byte offset

    const uint32_t wid = blockIdx.x * waves + wave, wtotal = gridDim.x * waves;
    uint32_t bcur = wid;                         // wave-uniform: the batch the wave takes next (its first: its own number)
    bool first = true;
    uint32_t wb = 0u, we = 0u;                   // wave-uniform: jobs [wb, we) of the current batch not handed out yet
    bool dry = false;                            // wave-uniform: no batch left
    bool have = false;                           // this lane works on a pair
    bool parked = false;                         // ... has reached its suffix y and waits for the wave's lanes still inside their x
    uint32_t waiting = 0u;                       // (EXC) ... stands before an exception site: rounds it has waited for company + 1
    bool flushing = false;                       // (EXC) wave-uniform: the lanes at exception sites are being served
    SnkFastLane L;
    L.s.lx = 0u; L.s.yoff = 0u; L.cur = 0u;

    for (;;) {
#ifdef SNK_STATS
        const unsigned long long stat_o0 = clock64();
        const bool stat_be = __any(have && L.cur + L.step > L.mfl1);      // this pass between two loop entries serves a block end
        const unsigned int stat_r0 = P.rounds;
#endif
        // ---- hand jobs to the lanes that have none ----
        bool need = lane_on && !have;
        while (!dry && __any(need)) {
            if (wb >= we) {
                // (sequences with exceptions) the lanes of a wave serve each other at the sites of the suffix they share: a lane
                // that has finished waits for the wave's batch to end instead of starting on the next batch's suffix alone
                // (measured at 336 rows x 1024: 100 IUPAC codes per Mbp 62 -> 74 % of the pure rate, the rest unchanged)
                if (EXC && __any(have)) break;
                if (G.queue) {           // dynamic: the next jobs of the launch, as many as the wave has chains
                    const uint32_t want = FAR ? lanes : G.batch - ((G.short_last && wave + 1u == waves) ? 1u : 0u);      // (as many as the wave has chains)
                    uint32_t b = 0xFFFFFFFFu;
                    // (FAR) near the end of the launch the LDS waves finish what is left sooner than a far chain would
                    if (lane == 0u && !(FAR && G.far_stop && *(volatile uint32_t *)G.queue + G.far_stop >= G.n_jobs))
                        b = atomicAdd(G.queue, want);
                    wb = (uint32_t)__shfl((int)b, 0);
                    if (wb >= G.n_jobs) { dry = true; break; }
                    we = wb + want < G.n_jobs ? wb + want : G.n_jobs;
                } else {
                    if (first) first = false;
                    else bcur += wtotal;
                    // (a workgroup takes `cwg` consecutive jobs at a time, wave w of it those from w * batch on: the same as batch
                    // number bcur when every wave has `batch` chains, and one column of 83 rows for 21 + 21 + 21 + 20 chains)
                    const uint32_t cwg = G.batch * waves - G.short_last;
                    const uint32_t mine_n = G.batch - ((G.short_last && wave + 1u == waves) ? 1u : 0u);
                    wb = (bcur / waves) * cwg + wave * G.batch;
                    if (wb >= G.n_jobs) { dry = true; break; }
                    we = wb + mine_n < G.n_jobs ? wb + mine_n : G.n_jobs;
                }
            }
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(need);
            const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            const uint32_t avail = we - wb;
            const bool take = need && rank < avail;
            SnkJob job; job.xi = 0; job.yi = -1; job.out_idx = 0; job.snap = 0;
            if (take) job = snk_fast_job(G, wb + rank);
            // cooperative table initialisation of every taking lane: prefix snapshot, or the all-zero start state
            for (unsigned long long tm = __builtin_amdgcn_ballot_w64(take); tm; tm &= tm - 1ull) {
                const uint32_t l = (uint32_t)__builtin_ctzll(tm);
                const int xi  = __shfl(job.xi, (int)l);
                const int snp = __shfl(job.snap, (int)l);
                const uint32_t lc = TRI ? (l >> 4) * 5u + (l & 15u) / 3u : SPEC ? l >> 1 : l;      // the chain of lane l
                uint8_t *dst = snk_lds8 + flut + (size_t)(wave * lanes + lc) * SNK_FCHAIN_B;
                const uint32_t spos = T.snap_pos[xi];
                const bool use = (snp == 0) && (spos != 0u);
                const uint32_t *src = T.snap_fast + (size_t)xi * SNK_FSLOTS;
                // offsets from the virtual base of the block before spos (k3 = -lx mod 4, see snk_fast_steady)
                const uint32_t k3 = (0u - T.len[xi]) & 3u;
                const uint32_t pvb = spos - 65536u - k3;
                if (FAR) {        // absolute positions; 0 = stream start (no snapshot) or dead (with one: spos >= 65536 is past it)
                    uint32_t *gdst = G.far_tab + (size_t)(far_wave0 + l) * SNK_FSLOTS;
                    for (uint32_t t = lane; t < SNK_FSLOTS; t += SNK_COOP(64u)) {
                        const uint32_t a0 = use ? src[t] : 0u;
                        gdst[t] = (a0 != 0u && a0 + 65536u >= spos) ? a0 : 0u;
                    }
                    continue;
                }
                for (uint32_t t = lane; t < SNK_FSLOTS / 2u; t += SNK_COOP(64u)) {
                    uint32_t v = k3 * 0x10001u;                       // stream start: every slot holds position 0
                    if (use) {
                        const uint32_t a0 = src[2u * t], a1 = src[2u * t + 1u];
                        uint32_t lo = (a0 != 0u && a0 + 65536u >= spos) ? a0 - pvb : 0u;   // previous block, else dead
                        uint32_t hi = (a1 != 0u && a1 + 65536u >= spos) ? a1 - pvb : 0u;
                        v = lo | (hi << 16);
                    }
                    ((uint32_t *)dst)[t] = v;
                }
                if (EXC) {
                    uint32_t *ov = T.ovf + ((size_t)(blockIdx.x * waves + wave) * lanes + lc) * 4096u;
                    const uint32_t *gsrc = T.snap_gen + (size_t)xi * 4096u;
                    // (the overflow table is indexed through lut_ovi: the other case's hashes first, by their compact numbers --
                    // what the table swaps of the other-case mode read and write with coalesced accesses)
                    for (uint32_t t = lane; t < 4096u; t += SNK_COOP(64u)) ov[T.lut_ovi[t]] = use ? gsrc[t] : 0u;
                }
                // no snapshot: position 0 counts as "written in this block"
                for (uint32_t t = lane; t < SNK_FBMWORDS; t += SNK_COOP(64u)) ((uint32_t *)(dst + SNK_FSLOTS * 2u))[t] = use ? 0u : 0xFFFFFFFFu;
            }
            if (take) {
                snk_fast_lane_init(L, T, job); have = true; need = false; parked = false; waiting = 0u;
                if (EXC) {
                    L.g.xb = T.bytes_arena + T.bytes_off[job.xi];
                    L.g.yb = job.yi >= 0 ? T.bytes_arena + T.bytes_off[job.yi] : T.zero_pad + SNK_PAD;
                    L.g.lx = L.s.lx;
                    L.fx = T.exc_off[job.xi] != 0xFFFFFFFFu ? T.exc_flags + T.exc_off[job.xi] : nullptr;
                    L.fy = (job.yi >= 0 && T.exc_off[job.yi] != 0xFFFFFFFFu) ? T.exc_flags + T.exc_off[job.yi] : nullptr;
                    L.ovf = T.ovf + ((size_t)(blockIdx.x * waves + wave) * lanes + cidx) * 4096u;
                    L.rx = L.fx ? T.exc_runs + 2u * T.exc_roff[job.xi] : nullptr;
                    L.ry = L.fy ? T.exc_runs + 2u * T.exc_roff[job.yi] : nullptr;
                    L.ri = 0u; L.ron_y = false;
                    L.xlim = 0u; L.site_lo = L.site_hi = 0u;
                    L.olo = L.olim = 0u; L.oscan = 0xFFFFFFFFu;
                    // the start state (position 0 in every slot, or x's prefix snapshot) may point at exceptions of x
                    L.mask_until = L.fx ? L.pos + 65536u : 0u;
                }
            }
            const uint32_t asked = (uint32_t)__builtin_popcountll(mask);
            wb += asked < avail ? asked : avail;
        }
        if (!__any(have)) {
#ifdef SNK_STATS
            if (lane == 0u) {
                unsigned long long *S = snk_stats + (FAR ? 32 : 0);
                atomicAdd(&S[7], clock64() - stat_w0); atomicAdd(&S[51], wall_clock64() - stat_wall0);
                atomicMax(&S[52], wall_clock64() - stat_wall0); atomicMax(&S[53], ~stat_wall0); atomicMax(&S[54], wall_clock64());   // longest wave, first start, last end
                atomicAdd(&S[13], P.loop); atomicAdd(&S[14], (unsigned long long)P.entries);
                atomicAdd(&S[24], P.finish); atomicAdd(&S[25], P.rounds_cyc); atomicAdd(&S[26], (unsigned long long)P.rounds);
                atomicAdd(&S[27], P.prologue); atomicAdd(&S[28], P.probe); atomicAdd(&S[29], P.top);
                atomicAdd(&S[30], (unsigned long long)P.jobs); atomicAdd(&S[31], 1ull); atomicAdd(&S[23], P.other); atomicAdd(&S[48], P.otrips); atomicAdd(&S[49], P.olanes); atomicAdd(&S[50], (unsigned long long)P.oruns);
                atomicAdd(&S[59], P.oin); atomicAdd(&S[60], P.oout); atomicAdd(&S[61], (unsigned long long)P.oswaps);
                atomicAdd(&S[62], P.be_cyc); atomicAdd(&S[63], (unsigned long long)P.be_n); atomicAdd(&S[55], (unsigned long long)P.be_rounds);
            }
#endif
            return;
        }

        // ---- (EXC) exception sites are served in company ----
        // The byte-accurate probes of a site cost a few thousand cycles each (cold ASCII lines), during which
        // the other lanes of the wave stand still.  The lanes of a wave walk the same y a few hundred trips
        // apart, so a lane that reaches a site waits (masked out of the steady loop) for the lanes behind it:
        // all waiting lanes are served together, in lockstep, once no running lane on the same y is less than
        // SNK_EXC_GATHER bases short of the foremost waiting one (or a lane waits in x, where the lanes do not
        // share the data, or the wait has lasted 64 rounds).  The sites thus pull the wave's band together
        // again; serving small groups as they came (round 2's first policy: 8 lanes, or 2 short rounds) spread it
        // further at every site -- 7 to 9 services per site and wave instead of 1 or 2 (measured, 1 Mbp genomes,
        // against pure ACGT: ten 100-base N runs 82 -> 89 %, 20 IUPAC codes 75 -> 86 %, 100 IUPAC codes 54 -> 72 %;
        // gather distance 1 Ki / 4 Ki / 16 Ki / 64 Ki bases: 62 / 55 / 72 / 72 % at 100 codes).
        if (EXC) {
            const unsigned long long wm = __builtin_amdgcn_ballot_w64(waiting != 0u);
            if (wm && !flushing) {
                // (the waiting lanes stand at one site as a rule: the first of them is the reference; a lane waiting elsewhere
                // only makes the wait a little shorter or longer)
                const int l0 = (int)__builtin_ctzll(wm);
                const uint32_t wfront = (uint32_t)__builtin_amdgcn_readlane((int)(L.cur - L.s.lx), l0);
                const uint32_t wy = (uint32_t)__builtin_amdgcn_readlane((int)L.s.yoff, l0);
                const bool in_x = __any(waiting != 0u && L.cur < L.s.lx);
                const bool runner = have && !parked && waiting == 0u && L.cur >= L.s.lx && L.s.yoff == wy;
                const uint32_t ry = L.cur - L.s.lx;
                const bool soon = runner && ry < wfront && wfront - ry < SNK_EXC_GATHER;
                if (in_x || !__any(soon) || __any(waiting > SNK_EXC_WAITMAX)) { flushing = true; if (lane == 0u) SNK_COUNT(3); }
            }
            if (flushing) waiting = 0u;
            else if (waiting) waiting++;
        }

        // ---- general probes and reservoir re-seats until every working lane is eligible ----
        bool refill = false;                     // wave-uniform: a lane has finished and a job may be left for it
#ifdef SNK_STATS
        const unsigned long long stat_i0 = clock64();
        P.top += stat_i0 - stat_o0;                      // job hand-out, gathering at sites
#if SNK_STATS + 0 >= 3
        if (stat_be && lane == 0u) atomicAdd(&snk_stats[32], stat_i0 - stat_o0);
#endif
#endif
        for (;;) {
#ifdef SNK_STATS
            P.rounds++;
#if SNK_STATS + 0 >= 3
            const unsigned long long stat_r_ = clock64();
#endif
#endif
            // Lanes of a wave share their suffix y (jobs are suffix-major) but reach it after x tails of
            // different lengths.  A lane that arrives parks until no lane of the wave is inside its x any
            // more: the wave then walks y as one band of a few KB and the L1 keeps serving the windows.
            const bool anyx = __any(have && L.cur < L.s.lx + 4u);      // wave-uniform: some lane is still inside its x
            if (!anyx) parked = false;
            if (EXC && have && L.cur >= L.xlim && L.cur + L.step <= L.mfl1)
                L.xlim = snk_exc_next(L, L.cur);             // (the cursor itself when its window is not clean)
            if (EXC && !flushing && waiting == 0u && have && L.cur + L.step <= L.mfl1 && L.cur >= L.xlim)
                { waiting = 1u; SNK_COUNT_HEAVY(5); }              // at an exception site: wait for company
            bool ok = !have || parked || waiting != 0u || snk_fast_eligible<EXC>(L);
            if (!ok && L.cur + L.step <= L.mfl1 && (!EXC || L.cur < L.xlim)) {   // inside a block (and not at an exception site): can the reservoir be re-seated?
                const uint32_t cur = L.cur, lx = L.s.lx;
                if (cur >= lx + 4u) {
                    if (L.w.org != lx && anyx) parked = true;                 // first time on y
                    snk_win_init(L.w, L.s.arena, L.s.yoff, lx, 0xFFFFFFFFu, cur);
                } else if (cur >= 4u && cur + 12u <= lx)   snk_win_init(L.w, L.s.arena, L.s.xoff, 0u, lx - 12u, cur);
                ok = parked || snk_fast_eligible<EXC>(L);
            }
            bool oel = false;
            if (EXC) {
                // lanes served at a site that stand inside an other-case stretch (a window of class 01 only) walk it in the steady
                // loop's other-case mode instead of one general probe per round -- TOGETHER: the lanes gathered at the site reach
                // the stretch proper a few general probes apart, so the mode starts once no served lane needs a general probe any
                // more (else the first lane to arrive would walk its whole stretch alone, then the next one, ...)
                oel = !ok && flut > SNK_FLUT_B && L.cur + L.step <= L.mfl1 && L.step == 1u && L.nb < 63u + SNK_FAST_MAXLIT &&
                      L.op + SNK_FAST_ZONE <= L.olimit && L.cur >= L.xlim && snk_exc_other_ready(L);
                if (__any(oel) && !__any(!ok && !oel)) {
#ifdef SNK_STATS
                    const unsigned long long stat_q0 = clock64();
#endif
                    // the chains of these lanes swap tables (every lane of the wave helps with every chain), walk their stretches in the
                    // other-case mode -- all together, re-entering after a service that leaves a lane inside its stretch with an
                    // ordinary next probe (finishing a probe touches no table) --, and swap back
                    const unsigned long long om_ = __builtin_amdgcn_ballot_w64(oel);
                    snk_oth_swap_in_all<SPEC>(T, snk_lds8, om_, flut, wave, lanes, (size_t)(blockIdx.x * waves + wave) * lanes, L.base, lane);
#ifdef SNK_STATS
                    const unsigned long long stat_q1 = clock64();
                    P.oin += stat_q1 - stat_q0; P.oswaps++;
#endif
                    bool inm = oel;
                    while (__any(inm)) {
                        if (inm) {
                            const uint32_t cur = L.cur, lx = L.s.lx;
                            if (cur >= lx + 4u) snk_win_init(L.w, L.s.arena, L.s.yoff, lx, 0xFFFFFFFFu, cur);
                            else                snk_win_init(L.w, L.s.arena, L.s.xoff, 0u, lx - 12u, cur);
                            if (L.mask_until < L.olim + 65536u) L.mask_until = L.olim + 65536u;     // its puts point into the stretch
                            snk_fast_steady<ASM, true, false, true>(L, (snk_g8 *)T.packed_arena, (snk_g8 *)T.mask_arena, tbl, bm, gt,
                                                                    (SNK_AS1 uint32_t *)G.far_tab, gtb, slot, mine_off, 0xFFFFFFFFu, T.lut_okey SNK_PROF_PASS);
                            inm = L.cur - L.olo < L.olim - L.olo && L.cur + L.step <= L.mfl1 && L.step == 1u &&
                                  L.nb < 63u + SNK_FAST_MAXLIT && L.op + SNK_FAST_ZONE <= L.olimit;
                        }
                    }
#ifdef SNK_STATS
                    const unsigned long long stat_q2 = clock64();
#endif
                    snk_oth_swap_out_all<SPEC>(T, snk_lds8, om_, flut, wave, lanes, (size_t)(blockIdx.x * waves + wave) * lanes, L.base, lane);
#ifdef SNK_STATS
                    P.other += clock64() - stat_q0; P.oout += clock64() - stat_q2;
#endif
                    continue;
                }
            }
            if (__builtin_expect(!__any(!ok), 1)) { flushing = false; break; }
#ifdef SNK_STATS
            const unsigned long long stat_g0 = clock64();
#if SNK_STATS + 0 >= 3
            if (stat_be && lane == 0u) { atomicAdd(&snk_stats[33], stat_g0 - stat_r_); atomicAdd(&snk_stats[36], 1ull); }
#endif
#endif
            bool stat_done = false;
            const bool edge = !ok && !oel && L.cur + L.step > L.mfl1;      // this lane closes a block / opens the next one in this round
            bool served = !ok && !oel;
            if (served && snk_fast_iter<EXC, FAR>(L, T, tbl, bm, gt, slot, out, status)) { have = false; stat_done = true; served = false; }   // frame complete
            if (EXC) {
                // A lane served at a site goes on with general probes HERE for as long as its cursor window holds a byte of another
                // class than the rest of it (the 16 cursors around the edge of a soft-masked stretch, an IUPAC code, a run of N):
                // a round of this loop costs the wave ~7 k cycles of which the probe is 2.8 k -- the site scan, the eligibility
                // tests, the other-case scan, the ballots are the rest -- and a stretch's two edges were ~27 rounds.  A general
                // probe is right at any position; the round's logic looks at the lane again when the window is of one class
                // (or after 24 probes).
                for (uint32_t n = 0; n < 24u; ++n) {
                    bool more = false;
                    if (served && have && L.cur + L.step <= L.mfl1 && (L.cur >= L.s.lx + 4u || (L.cur >= 4u && L.cur + 12u <= L.s.lx))) {
                        const uint32_t cw = snk_fetch32m(L.s, L.cur);
                        more = cw != 0u && cw != 0x55555555u;
                    }
                    if (!__any(more)) break;
                    if (more && snk_fast_iter<EXC, FAR>(L, T, tbl, bm, gt, slot, out, status)) { have = false; stat_done = true; served = false; }
                }
            }
            // (a block edge the block step has not seated the reservoir for -- the first block inside y, the seam --: seated here
            // instead of in a round of its own)
            if (!EXC && !FAR && edge && have && L.in_block && L.cur + L.step <= L.mfl1 && !snk_fast_eligible<EXC>(L)) {      // (else seated by the block step)
                const uint32_t cur = L.cur, lx = L.s.lx;
                if (cur >= lx + 4u) {
                    if (L.w.org != lx && anyx) parked = true;                 // first time on y
                    snk_win_init(L.w, L.s.arena, L.s.yoff, lx, 0xFFFFFFFFu, cur);
                } else if (cur >= 4u && cur + 12u <= lx)   snk_win_init(L.w, L.s.arena, L.s.xoff, 0u, lx - 12u, cur);
            }
            (void)stat_done;
#ifdef SNK_STATS
            P.jobs += (unsigned int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(stat_done));
            P.probe += clock64() - stat_g0;                                                 // inside the general probes
#if SNK_STATS + 0 >= 3
            const unsigned long long stat_g1 = clock64();
            if (stat_be && lane == 0u) atomicAdd(&snk_stats[34], stat_g1 - stat_g0);
#endif
#endif
            // (a finished lane goes back to the hand-out -- unless that would hold it anyway: sets with exceptions hand the next
            // batch out only when the wave's whole batch has ended, and a pass through the top of the loop for nothing re-runs
            // the ballots and the gathering logic and ages `waiting`)
            if (!dry && __any(lane_on && !have) && !(EXC && wb >= we && __any(have))) { refill = true; break; }
        }
#ifdef SNK_STATS
        P.rounds_cyc += clock64() - stat_i0;
        if (stat_be) { P.be_cyc += clock64() - stat_o0; P.be_n++; P.be_rounds += P.rounds - stat_r0; }      // (top + rounds; finish and prologue are per entry)
#endif
        if (refill) continue;
        if (!__any(have)) continue;              // the last working lane has just finished: hand out / leave
        uint32_t round = 0xFFFFFFFFu;
#if defined(SNK_PARK) && SNK_PARK == 1
        if (EXC) round = SNK_EXC_GATHER;                       // (lanes parked at a site are looked after at least this often)
#endif
#if defined(SNK_BAND) && SNK_BAND > 0
        // diagnostic build (sequences with exceptions): keep the wave's lanes that walk one suffix within SNK_BAND bases of the
        // hindmost of them, so that they reach an exception site within a few trips of each other
        if (EXC) {
            const bool iny = have && L.cur >= L.s.lx + 4u;
            const unsigned long long ym = __builtin_amdgcn_ballot_w64(iny);
            if (ym) {
                const uint32_t wy = (uint32_t)__builtin_amdgcn_readlane((int)L.s.yoff, (int)__builtin_ctzll(ym));
                const bool same = iny && L.s.yoff == wy;
                uint32_t yp = same ? L.cur - L.s.lx : 0xFFFFFFFFu;
                for (int o = 32; o; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)yp, o); yp = v < yp ? v : yp; }
                if (same) {
                    const uint32_t room = yp + (uint32_t)SNK_BAND - (L.cur - L.s.lx);      // bases this lane may still walk (>= SNK_BAND for the hindmost)
                    round = (int32_t)room <= 0 ? 0u : (room < round ? room : round);
                }
            }
        }
#endif
#ifdef SNK_HOST_EMU
        if (SPEC) {     // the emulated lane is role 0 of its pair; its partner lives for the duration of the loop (snk_host_emu.h)
            if (have && !parked && waiting == 0u && round != 0u)
                snk_emu_pair_run([&](bool r1, SnkFastLane &Lr) {
                    snk_fast_steady_spec<false, EXC>(Lr, r1, T, slot, out, status, tbl, bm, mine_off, round SNK_PROF_PASS);
                }, L);
        } else
#else
        if (TRI) {
            const uint32_t go = (have && !parked && round != 0u) ? 1u : 0u;      // (every lane of the wave is active here)
            const uint32_t g1 = snk_row_shr1(go), g2 = snk_row_shr2(go);
            if (role == 0u ? go != 0u : role == 1u ? g1 != 0u : role == 2u ? g2 != 0u : false)
                snk_fast_steady_spec3<TRI ? TRI : 10u>(L, role, (snk_g8 *)T.packed_arena, tbl, bm SNK_PROF_PASS);
        } else if (SPEC) {
            const bool go = have && !parked && waiting == 0u && round != 0u;
            const bool pgo = snk_pair_swap(go ? 1u : 0u) != 0u;                 // (every lane of the wave is active here)
            if (go || ((lane & 1u) && pgo))
                snk_fast_steady_spec<ASM, EXC>(L, (lane & 1u) != 0u, T, slot, out, status, tbl, bm, mine_off, round SNK_PROF_PASS);
        } else
#endif
        if (have && !parked && waiting == 0u && round != 0u)
            snk_fast_steady<ASM, EXC, FAR>(L, (snk_g8 *)T.packed_arena, (snk_g8 *)(EXC ? T.mask_arena : T.packed_arena), tbl, bm,
                                           gt, (SNK_AS1 uint32_t *)G.far_tab, gtb, slot, mine_off, round, T.lut_okey SNK_PROF_PASS);
    }
}

template <bool ASM, bool EXC>
__device__ __forceinline__ void snk_fast_kernel_body(const SnkTables &T, const SnkFastGrid &G,
                                                     uint32_t lanes, uint32_t *out, uint32_t *status)
{
#ifndef SNK_HOST_EMU
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
#endif
    for (uint32_t t = threadIdx.x; t < 512u; t += SNK_COOP(blockDim.x))
        ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
    if (EXC && G.flut > SNK_FLUT_B)               // the other case's LUT (the other-case mode of the steady loop)
        for (uint32_t t = threadIdx.x; t < 512u; t += SNK_COOP(blockDim.x))
            ((uint32_t *)snk_lds8)[512u + t] = ((const uint32_t *)T.lut_oj)[t];
    __syncthreads();
    if (!EXC && G.far_lanes != 0u && (threadIdx.x >> 6) >= G.lds_waves)
        snk_fast_wave<ASM, false, true>(T, G, G.far_lanes, out, status);
    else
        snk_fast_wave<ASM, EXC, false>(T, G, lanes, out, status);
}

#ifndef SNK_HOST_EMU
// phase B, one lane per chain (option fast_spec = 0; far chains)
__global__ void __launch_bounds__(512) snk_fast_one_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body<true, false>(T, G, lanes, out, status);
}

// the same with the C++ statement of the steady loop (option fast_asm = 0: cross-check of the
// hand-scheduled loop in the tests, A/B timing)
__global__ void __launch_bounds__(512) snk_fast_cxx_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body<false, false>(T, G, lanes, out, status);
}

// phase B: ordered pairs of pure ACGT sequences (the dominant kernel of the bench): two lanes per chain, snk_fast_steady_spec
__global__ void __launch_bounds__(512) snk_fast_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    for (uint32_t t = threadIdx.x; t < 512u; t += blockDim.x)
        ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
    __syncthreads();
    snk_fast_wave<true, false, false, true>(T, G, lanes, out, status);
}
// ... the C++ statement of that loop (fast_spec = 1, fast_asm = 0)
__global__ void __launch_bounds__(512) snk_fast_spec_cxx_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    for (uint32_t t = threadIdx.x; t < 512u; t += blockDim.x)
        ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
    __syncthreads();
    snk_fast_wave<false, false, false, true>(T, G, lanes, out, status);
}

// ... three lanes per chain, C++ statement (fast_spec = 3; round 4 experiment)
__global__ void __launch_bounds__(512) snk_fast_tri_cxx_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    for (uint32_t t = threadIdx.x; t < 512u; t += blockDim.x)
        ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
    __syncthreads();
    snk_fast_wave<false, false, false, false, 10u>(T, G, lanes, out, status);
}
// ... with role 2 six bases ahead instead of ten (fast_spec = 36)
__global__ void __launch_bounds__(512) snk_fast_tri6_cxx_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    for (uint32_t t = threadIdx.x; t < 512u; t += blockDim.x)
        ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
    __syncthreads();
    snk_fast_wave<false, false, false, false, 6u>(T, G, lanes, out, status);
}

// phase A: single sequences + prefix snapshots at upload (same code, own symbols so that profiles keep the two phases
// apart): two lanes per chain as in phase B (x-only blocks are what the DUAL form of the loop serves), one lane (fast_spec = 0)
__global__ void __launch_bounds__(512) snk_fast_singles_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    for (uint32_t t = threadIdx.x; t < 512u; t += blockDim.x)
        ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
    __syncthreads();
    snk_fast_wave<true, false, false, true>(T, G, lanes, out, status);
}
__global__ void __launch_bounds__(512) snk_fast_singles_one_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body<true, false>(T, G, lanes, out, status);
}

// the same again for resident sets in which some 2-bit sequence has exceptions (N runs, IUPAC codes): two lanes per chain
// (hand-scheduled / C++ statement), one lane per chain (fast_spec = 0), singles
template <bool ASM>
__device__ __forceinline__ void snk_fastx_spec_body(const SnkTables &T, const SnkFastGrid &G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    for (uint32_t t = threadIdx.x; t < 512u; t += blockDim.x) {
        ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
        if (G.flut > SNK_FLUT_B) ((uint32_t *)snk_lds8)[512u + t] = ((const uint32_t *)T.lut_oj)[t];         // the other case's LUT
    }
    __syncthreads();
    snk_fast_wave<ASM, true, false, true>(T, G, lanes, out, status);
}
__global__ void __launch_bounds__(512) snk_fastx_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fastx_spec_body<true>(T, G, lanes, out, status);
}
__global__ void __launch_bounds__(512) snk_fastx_spec_cxx_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fastx_spec_body<false>(T, G, lanes, out, status);
}
__global__ void __launch_bounds__(512) snk_fastx_one_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body<true, true>(T, G, lanes, out, status);
}
__global__ void __launch_bounds__(512) snk_fastx_cxx_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body<false, true>(T, G, lanes, out, status);
}
__global__ void __launch_bounds__(512) snk_fastx_singles_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fastx_spec_body<true>(T, G, lanes, out, status);
}
__global__ void __launch_bounds__(512) snk_fastx_singles_one_kernel(SnkTables T, SnkFastGrid G, uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_fast_kernel_body<true, true>(T, G, lanes, out, status);
}
#endif
