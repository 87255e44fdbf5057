// snk_bytes.hip.h -- tight-loop byte kernels: linked mode (full / compact tables) and one-shot mode.
// Part of the device code of libsnacc_hip.so; see snk_common.hip.h for the execution model.
#pragma once
#include "snk_common.hip.h"

#include "snk_fast.hip.h"     // snk_lit_ext, global-load helpers are in snk_common; nothing else is shared

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
    const uint16_t *glut;     // global-table kernels (no LDS at all): the hash -> slot LUT in global memory, for the slow paths
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
template <int CAP, bool ONESHOT, bool GT = false>
__device__ __forceinline__ uint32_t snk_bslot_slow(const SnkByteSrc &s, uint32_t p)
{
    const uint64_t w0 = snk_bld8(s, p);
    const uint32_t h = ONESHOT ? snk_hash4_u32((uint32_t)w0) : snk_hash5_parts((uint32_t)w0 << 24, (uint32_t)(w0 >> 8));
    if (CAP == 0) return h;
    const uint32_t id = GT ? (uint32_t)s.glut[h] : snk_bslot<true>(h);
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

template <int CAP, bool ONESHOT, bool GT = false>     // GT: `tbl` is the chain's table in global memory (u32 absolute positions, see snk_bytes_gt_body)
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
                const uint32_t t = (GT && CAP != 0) ? (uint32_t)L.s.glut[h] : snk_bslot<(CAP != 0)>(h);
                uint32_t v = 0u;
                if (CAP == 0 || t != SNK_BC_NOSLOT) {
                    if (GT) v = ((const uint32_t *)tbl)[t];              // the reader drops what is out of reach
                    else if ((bm[t >> 5] >> (t & 31u)) & 1u) v = L.pos - 65536u + tbl[t];
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
        if (!GT && !L.first) {
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
            const uint32_t s0 = snk_bslot_slow<CAP, ONESHOT, GT>(L.s, L.pos);
            if (GT) ((uint32_t *)tbl)[s0] = L.pos;
            else { tbl[s0] = 0; atomicOr(&bm[s0 >> 5], 1u << (s0 & 31u)); }
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
template <int CAP, bool ONESHOT, bool GT = false>
__device__ __forceinline__ void snk_bytes_table(const SnkByteLane &L, uint16_t *tbl, uint32_t *bm, uint32_t cur,
                                                uint32_t s1, uint32_t s2, uint32_t &cand, bool &valid)
{
    s2 = L.pending ? s2 : SnkBT<CAP, ONESHOT>::DUMMY;
    if (GT) {
        // liblz4's own table: absolute positions, a candidate counts while it is within 65535 bytes
        uint32_t *const gt = (uint32_t *)tbl;
        const uint32_t e = gt[s1];
        if (L.pending) gt[s2] = cur - 2u;          // nothing owed: no store at all (the kernel is bound by its memory traffic)
        gt[s1] = cur;
        const bool same = (s2 == s1);
        cand = same ? cur - 2u : e;
        valid = cur - cand <= 65535u;
        cand = valid ? cand : cur;
        return;
    }
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
template <int CAP, bool ONESHOT, bool GT = false>
__device__ __forceinline__ bool snk_bytes_iter_slow(SnkByteLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                                    uint32_t *out, uint32_t *status)
{
    const uint32_t cur = L.cur, next = cur + L.step;
    if (next > L.mfl1) return snk_bytes_block_step<CAP, ONESHOT, GT>(L, T, tbl, bm, out, status);
    const uint64_t wc = snk_bld8(L.s, cur);
    const uint32_t s1 = snk_bslot_slow<CAP, ONESHOT, GT>(L.s, cur);
    // the put of cur-2 is owed only after a match, which ends at least 5 positions into the block
    const uint32_t s2 = L.pending ? snk_bslot_slow<CAP, ONESHOT, GT>(L.s, cur - 2u) : SnkBT<CAP, ONESHOT>::DUMMY;
    uint32_t cand; bool valid;
    snk_bytes_table<CAP, ONESHOT, GT>(L, tbl, bm, cur, s1, s2, cand, valid);
    const uint32_t s3 = L.nb >> 6;
    const uint64_t wd = snk_bld8(L.s, cand);
    if (valid && (uint32_t)wc == (uint32_t)wd) {
        snk_bytes_match_slow(L, cur, cand, L.anchor, L.op);
    } else {
        L.cur = next; L.step = s3 ? s3 : 1u; L.nb++; L.pending = false;
    }
    // seat the reservoir for the tight loop when the new cursor allows it (slot-stream loop: only which source serves the cursor)
    const uint32_t nc = L.cur;
    if (!ONESHOT && T.slots) {
        if (nc >= L.s.lx + 4u)                     { L.w.soff = L.s.yoff; L.w.org = L.s.lx; L.w.lim = 0xFFFFFFFFu; }
        else if (nc >= 4u && nc + 8u <= L.s.lx)    { L.w.soff = L.s.xoff; L.w.org = 0u; L.w.lim = L.s.lx - 8u; }
        else                                       L.w.lim = 0u;
        return false;
    }
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


// Tight loop of the linked mode on the SLOT STREAM (T.slots): no hash arithmetic, no hash -> slot LUT and no register
// reservoir on the dependent chain.  The next probe's data -- the slots of cur-2 and cur (one 8-byte load from the slot
// stream) and the 12-byte compare window of the cursor (one load from the ASCII arena) -- are loaded directly as soon as
// the next cursor is known; both hit the L1 (the wave's chains walk the same suffix).  Cursors whose 12-byte window does
// not lie inside one sequence (stream start, seam) go through the general probe, as before.
struct __attribute__((packed)) SnkU64p { uint64_t v; };
template <int CAP, bool GT = false>
__device__ __forceinline__ void snk_bytes_loop2(SnkByteLane &L, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                                uint32_t *out, uint32_t *status)
{
    snk_g8 *const arena = L.s.arena;
    const SNK_AS1 uint16_t *const slots = (const SNK_AS1 uint16_t *)T.slots;
    for (;;) {
        uint32_t cur, next;
        for (;;) {
            cur = L.cur;
            next = cur + L.step;
            const bool pre = (next > L.mfl1) | (cur > L.w.lim) | (cur < L.w.org + 4u);     // w: which source serves the cursor (soff, org, lim)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pre) == 0ull, 1)) break;
            if (pre && snk_bytes_iter_slow<CAP, false, GT>(L, T, tbl, bm, out, status)) return;
        }
        // data of the first probe
        uint32_t so = L.w.soff + (cur - L.w.org);                                           // ASCII arena offset of the cursor
        // full table in global memory: the slot IS liblz4's hash, taken from the window (the kernel waits for memory, not
        // for its ALU: the multiplications cost less than the slot stream's line per probe)
        constexpr bool HS = GT && CAP == 0;
        uint64_t sv = 0;
        if (!HS) sv = ((const SNK_AS1 SnkU64p *)(slots + (size_t)so - 2u))->v;               // slots of cur-2 .. cur+1
        const SNK_AS1 SnkU96 *wp = (const SNK_AS1 SnkU96 *)(arena + (size_t)so - 4u);
        SnkW12 w; w.a = wp->a; w.b = wp->b; w.c = wp->c;
        const uint32_t olim6 = L.olimit - 6u;

        for (;;) {
            uint32_t cand; bool valid;
            const uint32_t s1 = HS ? snk_hash5_parts(w.b << 24, __builtin_amdgcn_alignbit(w.c, w.b, 8)) : (uint32_t)(sv >> 32) & 0xFFFFu;
            const uint32_t s2 = HS ? snk_hash5_parts((w.a << 8) & 0xFF000000u, __builtin_amdgcn_alignbit(w.b, w.a, 24)) : (uint32_t)sv & 0xFFFFu;
            snk_bytes_table<CAP, false, GT>(L, tbl, bm, cur, s1, s2, cand, valid);
            const SnkW12 wd = snk_bfetch12(L.s, cand);

            const uint32_t x0 = w.a ^ wd.a, x1 = w.b ^ wd.b, x2 = w.c ^ wd.c;
            uint32_t fh = (uint32_t)__builtin_ctz(x2 | 0x80000000u) >> 3;
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

            // ---- the next probe's data: direct loads (in flight during the bookkeeping) ----
            const uint32_t nso = so + (ncur - cur);
            uint64_t nsv = 0;
            if (!HS) nsv = ((const SNK_AS1 SnkU64p *)(slots + (size_t)nso - 2u))->v;
            const SNK_AS1 SnkU96 *nwp = (const SNK_AS1 SnkU96 *)(arena + (size_t)nso - 4u);
            SnkW12 nw; nw.a = nwp->a; nw.b = nwp->b; nw.c = nwp->c;

            // ---- bookkeeping of this probe ----
            const uint32_t anchor0 = L.anchor, op0 = L.op;
            uint32_t lit = cur - anchor0;
            uint32_t b = eq < lit ? eq : lit;
            b = b < cand ? b : cand;
            lit -= b;
            const uint32_t opn = op0 + lit + 3u;
            const bool rare = m & ((b == 4u) | (f == 8u) | (lit >= 15u) | (opn > olim6));
            const bool pre = (nnext > L.mfl1) | (ncur > L.w.lim);
            L.op = m ? opn : op0;
            L.anchor = m ? e2 : anchor0;
            L.step = nstep;
            L.nb = m ? 63u : L.nb + 1u;
            L.cur = ncur;
            L.pending = m;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare | pre) != 0ull, 0)) {
                if (rare) snk_bytes_match_slow(L, cur, cand, anchor0, op0);
                break;
            }
            cur = ncur; next = nnext; so = nso; sv = nsv; w = nw;
        }
    }
}

// ---- slot-stream loop with a speculative partner lane per chain (round 3, "bytes_spec") -----------------------------
// The LDS byte kernels wait for memory on every trip (candidate window from a 64 KiB history, then the next probe's
// slots + window) with at most 70 chains per CU and 17 of a wave's 64 lanes in use.  As in the 2-bit kernel
// (snk_fast_steady_spec) the idle lanes run the SAME chain ahead: every chain is a DPP pair (2i, 2i+1); lane 2i ("role 0")
// is the chain, lane 2i+1 ("role 1") probes at cur + 5 in the same trip exactly as liblz4's immediate probe after a
// 5-byte match would (put(cur + 3) owed, no literals, step 1).  Role 1 READS the table with everybody -- role 0's owed put
// has been issued before the read, its put(cur) and role 1's own owed put are patched into the result by selects, in
// liblz4's order -- and WRITES nothing until its probe is known to count: role 0's probe is an ordinary match that ends
// exactly where role 1 probed, and role 1's own probe needs no rare path and no limit handling (else it is dropped and
// role 0 makes that probe itself in the next trip: exactness never rests on the guess).  General probes, block steps
// and the rare paths are role 0's alone; role 1 takes the chain's state again at every entry of the tight loop.
__device__ __forceinline__ uint32_t snk_bpair_swap(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
template <int CAP>
__device__ __forceinline__ void snk_bytes_loop2_spec(SnkByteLane &L, const bool R1, const SnkTables &T, uint16_t *tbl, uint32_t *bm,
                                                     uint32_t *out, uint32_t *status)
{
    constexpr uint32_t DUMMY = SnkBT<CAP, false>::DUMMY;
    snk_g8 *const arena = L.s.arena;
    const SNK_AS1 uint16_t *const slots = (const SNK_AS1 uint16_t *)T.slots;
    bool done = false;
    for (;;) {
        // ---- role 0 brings its chain to a cursor the tight loop can serve; role 1 waits ----
        for (;;) {
            bool pre = false;
            if (!R1 && !done) {
                const uint32_t cur = L.cur, next = cur + L.step;
                pre = (next > L.mfl1) | (cur > L.w.lim) | (cur < L.w.org + 4u);
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pre) == 0ull, 1)) break;
            if (pre && snk_bytes_iter_slow<CAP, false, false>(L, T, tbl, bm, out, status)) done = true;
        }
        {   // a finished chain leaves with both its lanes
            const bool pdone = snk_bpair_swap(done ? 1u : 0u) != 0u;
            if (R1 ? pdone : done) return;
        }
        // ---- entry: role 1 takes the chain's state ----
        uint32_t cur = L.cur, step = L.step, nb = L.nb, anchor = L.anchor, op = L.op, pend = L.pending ? 1u : 0u;
        uint32_t base = L.base, mfl1 = L.mfl1, mlimit = L.mlimit, olimit = L.olimit, wsoff = L.w.soff, worg = L.w.org, wlim = L.w.lim;
        {
            const uint32_t a0 = snk_bpair_swap(cur), a1 = snk_bpair_swap(base), a2 = snk_bpair_swap(mfl1), a3 = snk_bpair_swap(mlimit);
            const uint32_t a4 = snk_bpair_swap(olimit), a5 = snk_bpair_swap(wsoff), a6 = snk_bpair_swap(worg), a7 = snk_bpair_swap(wlim);
            if (R1) { cur = a0 + 5u; base = a1; mfl1 = a2; mlimit = a3; olimit = a4; wsoff = a5; worg = a6; wlim = a7;
                      step = 1u; nb = 63u; anchor = cur; op = 0u; pend = 1u; }
        }
        const uint32_t olim6 = olimit - 6u;
        // this lane's slots (cur-2 .. cur+1) and 12-byte window; a role-1 cursor beyond the limits loads its partner's (never offered)
        uint32_t lc = (R1 && ((cur > mfl1) | (cur > wlim))) ? cur - 5u : cur;
        uint32_t so = wsoff + (lc - worg);
        uint64_t sv = ((const SNK_AS1 SnkU64p *)(slots + (size_t)so - 2u))->v;
        SnkW12 w;
        { const SNK_AS1 SnkU96 *wp = (const SNK_AS1 SnkU96 *)(arena + (size_t)so - 4u); w.a = wp->a; w.b = wp->b; w.c = wp->c; }

        for (;;) {
            const uint32_t s1 = (uint32_t)(sv >> 32) & 0xFFFFu;
            const uint32_t s2 = (R1 || pend) ? ((uint32_t)sv & 0xFFFFu) : DUMMY;
            const uint32_t c = cur - base;
            // ---- table, liblz4's order for the chain: role 0 put(cur-2) [owed], get(cur), put(cur); role 1 only reads ----
            const uint32_t b2 = R1 ? DUMMY : s2, b1 = R1 ? DUMMY : s1;
            const uint32_t bit1 = 1u << (s1 & 31u);
            tbl[b2] = (uint16_t)(c - 2u);
            atomicOr(&bm[b2 >> 5], 1u << (b2 & 31u));
            const uint32_t e = tbl[s1];
            const uint32_t bw = atomicOr(&bm[s1 >> 5], R1 ? 0u : bit1);
            tbl[b1] = (uint16_t)c;
            const bool iscur = (bw & bit1) != 0u;
            uint32_t cand = base + e - (iscur ? 0u : 65536u);
            bool valid = iscur | (e > c);
            {   // role 1: role 0's put(cur) of this trip and its own owed put have not been written
                const uint32_t ps1 = snk_bpair_swap(s1);
                if (R1 && ps1 == s1) { cand = cur - 5u; valid = true; }
                if (R1 && s2 == s1)  { cand = cur - 2u; valid = true; }
            }
            cand = valid ? cand : cur;
            const SnkW12 wd = snk_bfetch12(L.s, cand);

            const uint32_t x0 = w.a ^ wd.a, x1 = w.b ^ wd.b, x2 = w.c ^ wd.c;
            uint32_t fh = (uint32_t)__builtin_ctz(x2 | 0x80000000u) >> 3;
            fh = x2 ? fh : 4u;
            const uint32_t f = x1 ? ((uint32_t)__builtin_ctz(x1) >> 3) : 4u + fh;
            const uint32_t eq = x0 ? ((uint32_t)__builtin_clz(x0) >> 3) : 4u;
            const bool m = valid & (x1 == 0u);
            uint32_t e2 = cur + f;
            const bool clipped = e2 >= mlimit;
            e2 = e2 < mlimit ? e2 : mlimit;
            const uint32_t s3 = nb >> 6;
            const uint32_t nstep = m ? 1u : (s3 ? s3 : 1u);
            const uint32_t next = cur + step;
            const uint32_t ncur = m ? e2 : next;
            const uint32_t nnext = ncur + nstep;

            // ---- this probe's account (role 1: no literals, nothing before the cursor to catch up) ----
            uint32_t lit = cur - anchor;
            uint32_t b = eq < lit ? eq : lit;
            b = b < cand ? b : cand;
            lit -= b;
            const uint32_t opn = op + lit + 3u;
            const bool rare = m & ((b == 4u) | (f == 8u) | (lit >= 15u) | (opn > olim6));
            const bool pre = (nnext > mfl1) | (ncur > wlim);

            // ---- the pair decides ----
            // role 0 offers "an ordinary match that ends where role 1 probed, with room for role 1's sequence";
            // role 1 offers "a probe the tight loop may make, that needs neither a rare path nor limit handling"
            const bool offer = R1 ? ((next <= mfl1) & (cur <= wlim) & !rare & !pre)
                                  : (m & (f == 5u) & !clipped & !rare & !pre & (opn + 3u <= olim6));
            const uint32_t pk = (ncur & 0x3FFFFFFFu) | (m ? 0x40000000u : 0u) | (offer ? 0x80000000u : 0u);
            const uint32_t qk = snk_bpair_swap(pk);
            const bool com = offer & ((int32_t)qk < 0);                 // role 1's probe counts (the same value in both lanes)
            if (!R1) { SNK_COUNT(58); if (com) SNK_COUNT(59); if (offer) SNK_COUNT(60); if ((int32_t)qk < 0) SNK_COUNT(61); }
            if (R1 && com) {                                            // ... its two puts, after role 0's (liblz4's order)
                tbl[s2] = (uint16_t)(c - 2u);
                atomicOr(&bm[s2 >> 5], 1u << (s2 & 31u));
                tbl[s1] = (uint16_t)c;
                atomicOr(&bm[s1 >> 5], bit1);
            }
            const uint32_t n1 = R1 ? pk : qk;                           // (next cursor, matched) of role 1's probe
            const bool m1 = (n1 & 0x40000000u) != 0u;
            const uint32_t cp = com ? (n1 & 0x3FFFFFFFu) : (R1 ? (qk & 0x3FFFFFFFu) : ncur);   // the chain's next cursor

            // ---- commit (role 0 keeps the chain's account) ----
            const uint32_t anchor0 = anchor, op0 = op;
            if (!R1) {
                op = m ? opn : op0;
                anchor = m ? e2 : anchor0;
                step = nstep;
                nb = m ? 63u : nb + 1u;
                pend = m ? 1u : 0u;
                if (com) {                                              // role 1's probe: token + offset when it matched
                    op += m1 ? 3u : 0u;
                    anchor = m1 ? cp : anchor;
                    step = 1u;
                    nb = m1 ? 63u : 64u;
                    pend = m1 ? 1u : 0u;
                }
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!R1 & (rare | pre)) != 0ull, 0)) {
                if (!R1) {
                    L.op = op; L.anchor = anchor; L.step = step; L.nb = nb; L.cur = cp; L.pending = pend != 0u;
                    if (rare) snk_bytes_match_slow(L, cur, cand, anchor0, op0);
                }
                break;
            }
            // ---- the next probe's data ----
            const uint32_t ncl = R1 ? cp + 5u : cp;
            if (R1) anchor = ncl;
            lc = (R1 && ((ncl > mfl1) | (ncl > wlim))) ? cp : ncl;
            const uint32_t nso = wsoff + (lc - worg);
            sv = ((const SNK_AS1 SnkU64p *)(slots + (size_t)nso - 2u))->v;
            { const SNK_AS1 SnkU96 *nwp = (const SNK_AS1 SnkU96 *)(arena + (size_t)nso - 4u); w.a = nwp->a; w.b = nwp->b; w.c = nwp->c; }
            cur = ncl; so = nso;
        }
    }
}

// grid: one workgroup per `lanes*waves` jobs; dynamic LDS = LUT_B + CHAIN_B per chain.
// SPEC: two lanes per chain (snk_bytes_loop2_spec): chain l of a wave is the lane pair (2l, 2l+1); `lanes` stays the number
// of chains per wave (<= 32).
template <int CAP, bool ONESHOT, bool SPEC = false>
__device__ __forceinline__ void snk_bytes_kernel_body(const SnkTables &T, const SnkJob *jobs, uint32_t n_jobs,
                                                      uint32_t lanes, uint32_t *out, uint32_t *status)
{
    typedef SnkBT<CAP, ONESHOT> G;
    extern __shared__ __attribute__((aligned(16))) uint8_t snk_lds8[];
    const uint32_t tid = threadIdx.x, lane = SPEC ? (tid & 63u) >> 1 : tid & 63u, wave = tid >> 6;     // lane: the chain's number in its wave
    const bool R1 = SPEC && (tid & 1u);
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

    const uint32_t ilane = tid & 63u;                              // (the table fills below are by the wave's 64 lanes)
    for (uint32_t l = 0; l < lanes; ++l) {
        const int a   = __shfl((int)active, (int)(SPEC ? 2u * l : l));
        const int xi  = __shfl(job.xi, (int)(SPEC ? 2u * l : l));
        const int snp = __shfl(job.snap, (int)(SPEC ? 2u * l : l));
        if (!a) continue;
        uint8_t *dst = snk_lds8 + G::LUT_B + (size_t)(wave * lanes + l) * G::CHAIN_B;
        const uint32_t spos = T.snap_pos[xi];
        const bool use = !ONESHOT && (snp == 0) && (spos != 0u);
        const uint32_t *src = T.snap_gen + (size_t)xi * 4096u;
        if (CAP == 0) {
            for (uint32_t t = ilane; t < G::TBL_B / 4u; t += 64u) {
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
            for (uint32_t t = ilane; t < G::TBL_B / 4u; t += 64u) ((uint32_t *)dst)[t] = 0u;
            if (use) {
                // scatter liblz4's hash-indexed snapshot into the renamed slots (LDS ops of a wave are
                // executed in issue order, so the zero fill above lands first)
                for (uint32_t h = ilane; h < 4096u; h += 64u) {
                    const uint32_t id = snk_bslot<true>(h);
                    const uint32_t a0 = src[h];
                    if (id != SNK_BC_NOSLOT) ((uint16_t *)dst)[id] = (uint16_t)((a0 + 65536u >= spos) ? (a0 & 0xFFFFu) : 0u);
                }
            }
        }
        for (uint32_t t = ilane; t < G::BMWORDS; t += 64u)
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
    L.s.lx = lx; L.s.glut = nullptr;
    L.n = lx + ly;
    L.spos = T.snap_pos[job.xi];
    L.xi = job.xi; L.snap = job.snap; L.out_idx = job.out_idx;
    if (!ONESHOT && job.snap == 0 && L.spos != 0u) { L.pos = L.spos; L.total = T.snap_out[job.xi]; }
    else                                           { L.pos = 0u;     L.total = L.n ? T.header_bytes : 7u; }   // empty frame: no content-size field
    if (ONESHOT) L.snap = 0;
    L.blocks_left = (L.n >> 16) + 4u;
    L.iend = 0; L.blen = 0; L.first = true; L.in_block = false;
    L.cur = 0; L.step = 1; L.nb = 64; L.anchor = 0; L.op = 0;
    L.mfl1 = 0; L.mlimit = 0; L.olimit = 0; L.base = L.pos; L.endcode = 0;
    L.pending = false;
    L.w.soff = L.s.xoff; L.w.org = 0u; L.w.rb = 0u; L.w.lim = 0u;
    L.w.r0 = L.w.r1 = L.w.r2 = L.w.r3 = L.w.r4 = L.w.r5 = L.w.nx0 = L.w.nx1 = 0u;
    if (SPEC)                     snk_bytes_loop2_spec<CAP>(L, R1, T, tbl, bm, out, status);     // (launched only with a slot stream)
    else if (!ONESHOT && T.slots) snk_bytes_loop2<CAP>(L, T, tbl, bm, out, status);
    else                          snk_bytes_loop<CAP, ONESHOT>(L, T, tbl, bm, out, status);
}


// ---- byte kernel with the tables in global memory ("gt") ----------------------------------------------------------
// The LDS holds 18 full tables (70 compact ones) per CU, and the byte loop waits for memory on every trip (its
// candidate windows come from a 64 KiB history, twice the L1): with so few chains the CU idles.  This form keeps each
// chain's table -- liblz4's own layout, u32 absolute positions indexed by hash (or renamed slot), no bitmap, no ageing
// -- in global memory, so that every lane of every wave runs a chain (blockDim.x chains per workgroup): a trip costs
// one more trip to memory, and the CU has 30 times the chains to hide it behind.  The kernel uses no LDS at all (the
// slow paths read the hash -> slot LUT from global memory), so its workgroups fit beside any other kernel's.  The compact forms
// read the slot stream (T.slots); the full table takes liblz4's hashes from the window.  One launch runs at most `grid * blockDim.x` jobs (one table each in `gtab`); the host loops.
template <int CAP> struct SnkGT {
    static constexpr uint32_t NS = (SnkBT<CAP, false>::SLOTS + 1u + 15u) & ~15u;     // u32 entries per chain (incl. the dummy)
};
template <int CAP>
__device__ __forceinline__ void snk_bytes_gt_body(const SnkTables &T, const SnkJob *jobs, uint32_t n_jobs,
                                                  uint32_t *gtab, uint32_t *out, uint32_t *status)
{
    constexpr uint32_t NS = SnkGT<CAP>::NS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t c = blockIdx.x * blockDim.x + tid;
    const bool active = c < n_jobs;
    SnkJob job; job.xi = 0; job.yi = -1; job.out_idx = 0; job.snap = 0;
    if (active) job = jobs[c];
    uint32_t *const wave_tab = gtab + (size_t)(c - lane) * NS;

    for (uint32_t l = 0; l < 64u; ++l) {
        const int a   = __shfl((int)active, (int)l);
        const int xi  = __shfl(job.xi, (int)l);
        const int snp = __shfl(job.snap, (int)l);
        if (!a) continue;
        uint32_t *dst = wave_tab + (size_t)l * NS;
        const uint32_t spos = T.snap_pos[xi];
        const bool use = (snp == 0) && (spos != 0u);
        const uint32_t *src = T.snap_gen + (size_t)xi * 4096u;
        if (CAP == 0) {
            for (uint32_t t = lane; t < NS; t += 64u) dst[t] = (use && t < 4096u) ? src[t] : 0u;
        } else {
            for (uint32_t t = lane; t < NS; t += 64u) dst[t] = 0u;
            if (use) {
                __threadfence();                                  // the zero fill lands before the scatter
                for (uint32_t h = lane; h < 4096u; h += 64u) {
                    const uint32_t id = T.lut_h2c[h];
                    if (id != SNK_BC_NOSLOT) dst[id] = src[h];
                }
            }
        }
    }
    __threadfence();                                              // the tables are written before any lane reads its own
    if (!active) return;

    uint16_t *tbl = (uint16_t *)(wave_tab + (size_t)lane * NS);
    SnkByteLane L;
    const uint32_t lx = T.len[job.xi];
    const uint32_t ly = job.yi >= 0 ? T.len[job.yi] : 0u;
    L.s.arena = (snk_g8 *)T.bytes_arena;
    L.s.xoff = T.bytes_off[job.xi];
    L.s.yoff = job.yi >= 0 ? T.bytes_off[job.yi] : 16u;
    L.s.lx = lx;
    L.s.glut = T.lut_h2c;
    L.n = lx + ly;
    L.spos = T.snap_pos[job.xi];
    L.xi = job.xi; L.snap = job.snap; L.out_idx = job.out_idx;
    if (job.snap == 0 && L.spos != 0u) { L.pos = L.spos; L.total = T.snap_out[job.xi]; }
    else                               { L.pos = 0u;     L.total = L.n ? T.header_bytes : 7u; }
    L.blocks_left = (L.n >> 16) + 4u;
    L.iend = 0; L.blen = 0; L.first = true; L.in_block = false;
    L.cur = 0; L.step = 1; L.nb = 64; L.anchor = 0; L.op = 0;
    L.mfl1 = 0; L.mlimit = 0; L.olimit = 0; L.base = L.pos; L.endcode = 0;
    L.pending = false;
    L.w.soff = L.s.xoff; L.w.org = 0u; L.w.rb = 0u; L.w.lim = 0u;
    L.w.r0 = L.w.r1 = L.w.r2 = L.w.r3 = L.w.r4 = L.w.r5 = L.w.nx0 = L.w.nx1 = 0u;
    snk_bytes_loop2<CAP, true>(L, T, tbl, nullptr, out, status);
}

__global__ void __launch_bounds__(512) snk_bytes_gt_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                                           uint32_t *gtab, uint32_t *out, uint32_t *status)
{
    snk_bytes_gt_body<0>(T, jobs, n_jobs, gtab, out, status);
}
__global__ void __launch_bounds__(512) snk_bytes_gt1k_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                                             uint32_t *gtab, uint32_t *out, uint32_t *status)
{
    snk_bytes_gt_body<1024>(T, jobs, n_jobs, gtab, out, status);
}
__global__ void __launch_bounds__(512) snk_bytes_gt2k_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                                             uint32_t *gtab, uint32_t *out, uint32_t *status)
{
    snk_bytes_gt_body<2048>(T, jobs, n_jobs, gtab, out, status);
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

// two lanes per chain (linked mode on the slot stream; "bytes_spec")
__global__ void snk_bytes_spec_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                      uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<0, false, true>(T, jobs, n_jobs, lanes, out, status);
}
__global__ void snk_bytes_compact_spec_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                              uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<1024, false, true>(T, jobs, n_jobs, lanes, out, status);
}
__global__ void snk_bytes_compact2k_spec_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                                uint32_t lanes, uint32_t *out, uint32_t *status)
{
    snk_bytes_kernel_body<2048, false, true>(T, jobs, n_jobs, lanes, out, status);
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
