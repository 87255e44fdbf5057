// snk_deflate.hip.h -- device code of the gzip / zlib path: index kernels, the
// wave-per-job deflate_slow parser with zlib's block pricing, restart records, segment stitch.
// Included by snk_deflate.hip only; see that file for the overview.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr uint32_t DFL_MAX_DIST = 32506u;      // w_size - MIN_LOOKAHEAD
constexpr uint32_t DFL_MAX_MATCH = 258u;
constexpr uint32_t DFL_MIN_LOOKAHEAD = 262u;
constexpr uint32_t DFL_TOO_FAR = 4096u;
constexpr uint32_t DFL_BLOCK_SYMS = 16383u;    // lit_bufsize - 1
constexpr uint32_t DFL_NHASH = 32768u;
constexpr uint32_t DFL_RESTART_BACK = 600u;    // restart this far before the seam (> 258 + 262)
constexpr uint32_t DFL_HIST = 320u;            // 0..285 literal/length codes, 288..317 distance codes
constexpr uint32_t DFL_DOFF = 288u;
constexpr uint32_t DFL_HEAP = 573u;            // 2 * L_CODES + 1
constexpr uint32_t DFL_WAVES = 4u;             // wavefronts per workgroup
#ifndef DFL_WIDE_K
#define DFL_WIDE_K true                           // gzip kernel: the flush call's wide form (see dfl_flush)
#endif
#ifndef DFL_KW
#define DFL_KW 16u                               // bucket members the K-pass looks at first (then 64 at a time)
#endif

// LDS of a workgroup (bytes): per wavefront its symbol histogram; ONE set of the LDS tree-building arrays, used
// only by blocks with more than 64 distinct literal/length symbols (binary data) and taken under a lock -- DNA
// and protein blocks build their trees in registers (dfl_tree_small).
// The saved code lengths live in the wave's histogram memory: by the time they are written the counts they
// overwrite (literal codes 0..79) have been copied out, and the histogram is rebuilt afterwards.
constexpr uint32_t L_HIST = 0;                         // u32[320]
constexpr uint32_t L_LLEN = L_HIST;                    // u8[288]  code lengths of the literal/length tree
constexpr uint32_t L_DLEN = L_HIST + 288;              // u8[32]   code lengths of the distance tree
constexpr uint32_t L_BLF = L_HIST + 4 * DFL_HIST;      // u16[20]  bit-length code counts (scan_tree)
constexpr uint32_t L_MISC = L_BLF + 40;                // u32[6]   results of lane 0
constexpr uint32_t L_WAVE = L_MISC + 24;               // 1344 per wavefront
constexpr uint32_t G_HEAP = DFL_WAVES * L_WAVE;        // u16[576]   shared from here on
constexpr uint32_t G_FREQ = G_HEAP + 2 * 576;          // u16[576]
constexpr uint32_t G_DAD = G_FREQ + 2 * 576;           // u16[576]
constexpr uint32_t G_LEN = G_DAD + 2 * 576;            // u16[576]
constexpr uint32_t G_DEPTH = G_LEN + 2 * 576;          // u8[576]
constexpr uint32_t G_LOCK = G_DEPTH + 576;             // u32
constexpr uint32_t L_GROUP = G_LOCK + 16;              // 10 576 B per workgroup

struct DflSeq {              // one per resident sequence (device + host mirror)
    uint32_t boff, len;
    uint64_t ioff;           // element offset into occ / inv
    uint64_t soff;           // element offset into sym / pos (capacity len + 1)
    uint64_t coff;           // element offset into cumbits (capacity len / 16383 + 2)
    uint32_t nsym;           // symbols of the stand-alone stream
    uint32_t unsafe;         // 1: a stored-block decision near the end depends on the total length
    uint64_t total_bits;
    uint64_t koff;           // element offset into chk (checkpoint c of this sequence at koff + 320 * c)
    uint32_t rk, rpos;       // restart point: symbol index (clean state) and its stream position
    uint32_t rkb, rbpos;     // first symbol / stream position of the block that is open at rk
};

// mode 0: pair (or single, yi = -1) size from the stored streams;  1: whole sequence, serial (stores its stream,
// prices it);  2: segment [p0, p1) of a sequence into the scratch stream at `aux` (no pricing);  3: price the
// stored stream of a sequence
struct DflJob { int32_t xi, yi; uint32_t mode, out_idx; uint32_t p0, p1; uint64_t aux; };
constexpr uint32_t DFL_SEG = 32768u;           // segment length of the parallel per-sequence pass
constexpr uint32_t DFL_SEG_SLACK = 2048u;      // a segment runs this far into the next one, for the stitch
constexpr uint32_t DFL_ST_CAP = 2u;           // status bit: a stored symbol stream ran beyond its capacity
constexpr uint32_t DFL_SEG_ROOM = 300u;        // scratch entries beyond that (one match can carry the parser 258 further)

// Member of a six-byte bucket: everything the K-pass reads about it, in one 16-byte load (three parallel arrays cost
// three cache lines per group of candidates; the kernel is bound by the number of distinct lines it touches).
struct __attribute__((aligned(16))) DflKRec {
    uint64_t d8;              // the 8 bytes at the position
    uint32_t v;               // the position
    uint32_t r3;              // its rank in its 3-byte bucket
};

struct DflTables {
    const uint8_t *bytes;
    DflSeq *seq;
    const uint32_t *occ, *bstart;            // bstart: 32769 entries per sequence
    const uint64_t *occ8;                    // the 8 sequence bytes at occ[i] (coalesced candidate compares)
    const uint64_t *inv2;                    // per position: index in occ (low 32) | rank in its bucket (high 32)
    // second index, bucketed by a 16-bit hash of SIX bytes (positions <= len - 6): every chain member that
    // matches the probe in >= 6 bytes is in the probe's bucket
    const DflKRec *k_rec;                     // by six-byte bucket, positions ascending: one 16-byte record per member
    const uint32_t *k_bstart;                 // 65537 bucket bounds per sequence
    const uint64_t *k_inv2;                   // per position: index in k_rec | rank in its bucket << 32
    uint32_t use_k;
    uint32_t norestart;                       // option "deflate_norestart": pair jobs parse x from its start (testing)
    uint32_t *sym, *pos;
    uint64_t *cumbits;
    uint32_t *rhist;                         // 320 counters per sequence: open block at the restart point
    uint32_t *seg_sym, *seg_pos, *seg_cnt;   // scratch streams of the segment jobs, and their symbol counts
    const uint32_t *chk;                     // per sequence, per 128 symbols: counts of every code in the symbols before (320 each)
    uint32_t good, lazy, nice, chain;
    uint32_t *status;
};

// ---------------------------------------------------------------------------------------------
// index build
// ---------------------------------------------------------------------------------------------
__global__ void dfl_hash_kernel(const uint8_t *b, uint32_t m, uint16_t *key, uint32_t *val)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; p < m; p += stride) {
        key[p] = (uint16_t)((((uint32_t)b[p] << 10) ^ ((uint32_t)b[p + 1] << 5) ^ (uint32_t)b[p + 2]) & 0x7fffu);
        val[p] = p;
    }
}

__global__ void dfl_bstart_kernel(const uint16_t *skey, uint32_t m, uint32_t *bstart)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > DFL_NHASH) return;
    uint32_t lo = 0, hi = m;                  // first index with key >= h
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)skey[mid] < h) lo = mid + 1; else hi = mid;
    }
    bstart[h] = lo;
}

// inv2[p] = (index of p in occ) | (rank of p inside its hash bucket) << 32;  occ8[i] = the 8 bytes at occ[i]
__global__ void dfl_inv_kernel(const uint32_t *occ, const uint16_t *skey, const uint32_t *bstart, const uint8_t *b,
                               uint32_t m, uint64_t *inv2, uint64_t *occ8)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; i < m; i += stride) {
        const uint32_t p = occ[i];
        inv2[p] = (uint64_t)i | ((uint64_t)(i - bstart[skey[i]]) << 32);
        struct __attribute__((packed)) U64 { uint64_t v; };
        occ8[i] = ((const U64 *)(b + p))->v;               // the arena is zero padded behind every sequence
    }
}

__host__ __device__ __forceinline__ uint32_t dfl_hash6(uint64_t v8)     // 16-bit hash of the low 6 bytes
{
    return (uint32_t)(((v8 & 0xFFFFFFFFFFFFull) * 0x9E3779B97F4A7C15ull) >> 48);
}

__global__ void dfl_khash_kernel(const uint8_t *b, uint32_t m, uint16_t *key, uint32_t *val)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    struct __attribute__((packed)) U64 { uint64_t v; };
    for (; p < m; p += stride) { key[p] = (uint16_t)dfl_hash6(((const U64 *)(b + p))->v); val[p] = p; }
}

__global__ void dfl_kbstart_kernel(const uint16_t *skey, uint32_t m, uint32_t *bstart)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > 65536u) return;
    uint32_t lo = 0, hi = m;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)skey[mid] < h) lo = mid + 1; else hi = mid; }
    bstart[h] = lo;
}

__global__ void dfl_kinv_kernel(const uint32_t *kocc, const uint16_t *skey, const uint32_t *kbstart, const uint8_t *b,
                                const uint64_t *inv2, uint32_t m, uint64_t *kinv2, DflKRec *krec)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    struct __attribute__((packed)) U64 { uint64_t v; };
    for (; i < m; i += stride) {
        const uint32_t p = kocc[i];
        kinv2[p] = (uint64_t)i | ((uint64_t)(i - kbstart[skey[i]]) << 32);
        DflKRec r;
        r.d8 = ((const U64 *)(b + p))->v; r.v = p; r.r3 = (uint32_t)(inv2[p] >> 32);
        krec[i] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// the virtual stream x + y (+ what zlib's window holds behind the end of the input)
// ---------------------------------------------------------------------------------------------
struct DflStream {
    const uint8_t *X, *Y;
    uint32_t lx, ly, n;
    bool slid;                // zlib's window has slid at least once at the current loop top
};

// Behind the end of the input zlib's compare loop reads whatever the window buffer still holds:
// zeros while the window has not slid, else the bytes 32 KiB earlier (the buffer keeps its old
// content after a slide).  The first slide happens at the first loop top >= 65 274 with fewer than
// 262 bytes left -- also in a stream of 65 275 .. 65 536 bytes that fitted the buffer (s.slid).
__device__ __forceinline__ uint32_t dfl_byte(const DflStream &s, uint32_t a)
{
    if (a >= s.n) {
        if (!s.slid) return 0u;
        a -= 32768u;
    }
    return a < s.lx ? s.X[a] : s.Y[a - s.lx];
}

__device__ __forceinline__ uint64_t dfl_ld8(const uint8_t *p)
{
    struct __attribute__((packed)) U64 { uint64_t v; };
    return ((const U64 *)p)->v;
}

__device__ __forceinline__ uint64_t dfl_load8(const DflStream &s, uint32_t a)
{
    if (a + 8u <= s.lx) return dfl_ld8(s.X + a);
    if (a >= s.lx && a + 8u <= s.n) return dfl_ld8(s.Y + (a - s.lx));
    uint64_t v = 0;
    for (uint32_t i = 0; i < 8u; ++i) v |= (uint64_t)dfl_byte(s, a + i) << (8u * i);
    return v;
}

__device__ __forceinline__ uint32_t dfl_hash3(const DflStream &s, uint32_t a)
{
    return ((dfl_byte(s, a) << 10) ^ (dfl_byte(s, a + 1u) << 5) ^ dfl_byte(s, a + 2u)) & 0x7fffu;
}

// common prefix of the stream at p and at q < p, capped at 258
__device__ __forceinline__ uint32_t dfl_lcp(const DflStream &s, uint32_t p, uint32_t q)
{
    uint32_t t = 0;
    while (t < DFL_MAX_MATCH) {
        const uint64_t w = dfl_load8(s, p + t) ^ dfl_load8(s, q + t);
        if (w) { t += (uint32_t)__builtin_ctzll(w) >> 3; break; }
        t += 8u;
    }
    return t < DFL_MAX_MATCH ? t : DFL_MAX_MATCH;
}

// ---------------------------------------------------------------------------------------------
// trees.c: code lengths and block cost, by lane 0 of the wave, all arrays in the wave's LDS
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dfl_lcode(uint32_t lc)          // lc = match length - 3, 0..255
{
    if (lc < 8u) return lc;
    if (lc == 255u) return 28u;
    const uint32_t k = 31u - (uint32_t)__builtin_clz(lc);
    return 4u * k - 4u + ((lc >> (k - 2u)) & 3u);
}
__device__ __forceinline__ uint32_t dfl_dcode(uint32_t d)           // d = distance - 1, 0..32767
{
    if (d < 4u) return d;
    const uint32_t k = 31u - (uint32_t)__builtin_clz(d);
    return 2u * k + ((d >> (k - 1u)) & 1u);
}
__device__ __forceinline__ uint32_t dfl_lextra(uint32_t code)       // code 0..28 (length codes)
{
    return (code < 8u || code == 28u) ? 0u : (code >> 2) - 1u;
}
__device__ __forceinline__ uint32_t dfl_dextra(uint32_t code) { return code < 4u ? 0u : (code >> 1) - 1u; }
__device__ __forceinline__ uint32_t dfl_static_llen(uint32_t n) { return n <= 143u ? 8u : (n <= 255u ? 9u : (n <= 279u ? 7u : 8u)); }

struct DflLds {
    uint32_t *hist, *lock;
    uint16_t *blf;            // per wave: bit-length code counts of the small-alphabet path
    uint16_t *heap, *freq, *dad, *len;
    uint8_t *depth, *llen, *dlen;
    uint32_t *misc;
};

#define DFL_SMALLER(n, m) (L.freq[n] < L.freq[m] || (L.freq[n] == L.freq[m] && L.depth[n] <= L.depth[m]))

__device__ __forceinline__ void dfl_pqdownheap(const DflLds &L, int heap_len, int k)
{
    const int v = L.heap[k];
    int j = k << 1;
    while (j <= heap_len) {
        if (j < heap_len && DFL_SMALLER(L.heap[j + 1], L.heap[j])) j++;
        if (DFL_SMALLER(v, L.heap[j])) break;
        L.heap[k] = L.heap[j];
        k = j;
        j <<= 1;
    }
    L.heap[k] = (uint16_t)v;
}

// kind 0: literal/length tree, 1: distance tree, 2: bit-length tree.  Frequencies are in L.freq[0..elems).
// Leaves lengths in L.len[0..elems); returns max_code; adds to opt_len / static_len.
__device__ int dfl_build_tree(const DflLds &L, int kind, long &opt_len, long &static_len)
{
    const int elems = kind == 0 ? 286 : (kind == 1 ? 30 : 19);
    const int max_length = kind == 2 ? 7 : 15;
    int n, m, max_code = -1, node, h, bits, overflow = 0, heap_len = 0, heap_max = (int)DFL_HEAP;
    uint16_t bl_count[16];

    for (n = 0; n < elems; n++) {
        if (L.freq[n] != 0) { L.heap[++heap_len] = (uint16_t)(max_code = n); L.depth[n] = 0; }
        else L.len[n] = 0;
    }
    while (heap_len < 2) {
        node = (max_code < 2 ? ++max_code : 0);
        L.heap[++heap_len] = (uint16_t)node;
        L.freq[node] = 1;
        L.depth[node] = 0;
        opt_len--;
        if (kind == 0) static_len -= (long)dfl_static_llen((uint32_t)node);
        else if (kind == 1) static_len -= 5;
    }
    for (n = heap_len / 2; n >= 1; n--) dfl_pqdownheap(L, heap_len, n);
    node = elems;
    do {
        n = L.heap[1];
        L.heap[1] = L.heap[heap_len--];
        dfl_pqdownheap(L, heap_len, 1);
        m = L.heap[1];
        L.heap[--heap_max] = (uint16_t)n;
        L.heap[--heap_max] = (uint16_t)m;
        L.freq[node] = (uint16_t)(L.freq[n] + L.freq[m]);
        L.depth[node] = (uint8_t)((L.depth[n] >= L.depth[m] ? L.depth[n] : L.depth[m]) + 1);
        L.dad[n] = L.dad[m] = (uint16_t)node;
        L.heap[1] = (uint16_t)node++;
        dfl_pqdownheap(L, heap_len, 1);
    } while (heap_len >= 2);
    L.heap[--heap_max] = L.heap[1];

    for (bits = 0; bits < 16; bits++) bl_count[bits] = 0;
    L.len[L.heap[heap_max]] = 0;
    for (h = heap_max + 1; h < (int)DFL_HEAP; h++) {
        n = L.heap[h];
        bits = L.len[L.dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        L.len[n] = (uint16_t)bits;
        if (n > max_code) continue;
        bl_count[bits]++;
        int xbits = 0, slen = 0;
        if (kind == 0) { if (n >= 257) xbits = (int)dfl_lextra((uint32_t)n - 257u); slen = (int)dfl_static_llen((uint32_t)n); }
        else if (kind == 1) { xbits = (int)dfl_dextra((uint32_t)n); slen = 5; }
        else xbits = n == 16 ? 2 : (n == 17 ? 3 : (n == 18 ? 7 : 0));
        opt_len += (long)L.freq[n] * (bits + xbits);
        if (kind != 2) static_len += (long)L.freq[n] * (slen + xbits);
    }
    if (overflow > 0) {
        do {
            bits = max_length - 1;
            while (bl_count[bits] == 0) bits--;
            bl_count[bits]--;
            bl_count[bits + 1] += 2;
            bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (bits = max_length; bits != 0; bits--) {
            n = bl_count[bits];
            while (n != 0) {
                m = L.heap[--h];
                if (m > max_code) continue;
                if (L.len[m] != (uint16_t)bits) {
                    opt_len += ((long)bits - (long)L.len[m]) * (long)L.freq[m];
                    L.len[m] = (uint16_t)bits;
                }
                n--;
            }
        }
    }
    return max_code;
}

// scan_tree over saved code lengths; adds to the bit-length frequencies bl[0..19)
__device__ void dfl_scan_tree(const uint8_t *lens, int max_code, uint16_t *bl)
{
    int prevlen = -1, curlen, nextlen = lens[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen;
        nextlen = n == max_code ? 0xffff : lens[n + 1];          // zlib's guard entry
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) bl[curlen] = (uint16_t)(bl[curlen] + count);
        else if (curlen != 0) { if (curlen != prevlen) bl[curlen]++; bl[16]++; }
        else if (count <= 10) bl[17]++;
        else bl[18]++;
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

// opt_len / static_len of the block whose symbol counts are in L.hist (lane 0 only)
__device__ void dfl_block_lengths(const DflLds &L, long &opt_len, long &static_len)
{
    const uint8_t bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    opt_len = 0; static_len = 0;
    for (int n = 0; n < 286; n++) L.freq[n] = (uint16_t)L.hist[n];
    const int lmax = dfl_build_tree(L, 0, opt_len, static_len);
    for (int n = 0; n < 286; n++) L.llen[n] = (uint8_t)L.len[n];
    for (int n = 0; n < 30; n++) L.freq[n] = (uint16_t)L.hist[DFL_DOFF + n];
    const int dmax = dfl_build_tree(L, 1, opt_len, static_len);
    for (int n = 0; n < 30; n++) L.dlen[n] = (uint8_t)L.len[n];
    for (int n = 0; n < 19; n++) L.freq[n] = 0;
    dfl_scan_tree(L.llen, lmax, L.freq);
    dfl_scan_tree(L.dlen, dmax, L.freq);
    dfl_build_tree(L, 2, opt_len, static_len);
    int max_blindex;
    for (max_blindex = 18; max_blindex >= 3; max_blindex--)
        if (L.len[bl_order[max_blindex]] != 0) break;
    opt_len += 3 * ((long)max_blindex + 1) + 5 + 5 + 4;
}

// ---- the same tree construction for SMALL alphabets (<= 64 used symbols: any DNA or protein block), on arrays
// held in VGPR lanes: entry i of a "lane array" is lane i of `lo` (i < 64) or lane i - 64 of `hi`; all indices are
// wave-uniform, every access is one v_readlane / v_writelane, the control flow is scalar and nothing waits for LDS.
// Leaves are slots 0..m-1 in zlib's heap order (ascending code, forced ones last), internal nodes m..2m-2.
// zlib's smaller(n, m) -- lower frequency, or equal frequency and depth[n] <= depth[m] -- is key[n] <= key[m]
// with key = frequency << 8 | depth.
struct DflLA { uint32_t lo, hi; };
__device__ __forceinline__ uint32_t la_get(const DflLA &a, uint32_t i)
{
    return i < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)a.lo, (int)i)
                   : (uint32_t)__builtin_amdgcn_readlane((int)a.hi, (int)(i - 64u));
}
// (this clang has no writelane builtin; the s_nop covers the "VALU wrote the lane-select SGPR" hazard, which the
// compiler does not track into inline assembly)
__device__ __forceinline__ uint32_t dfl_writelane(uint32_t old, uint32_t val, uint32_t lane)
{
    const uint32_t sv = (uint32_t)__builtin_amdgcn_readfirstlane((int)val);     // both are wave-uniform already
    const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lane);
    // one SGPR operand per VALU instruction on this target: the lane select goes through M0
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(sv), "s"(sl));   // (M0 is reserved: the compiler sets it right before each of its own uses)
    return old;
}
__device__ __forceinline__ void la_set(DflLA &a, uint32_t i, uint32_t v)
{
    if (i < 64u) a.lo = dfl_writelane(a.lo, v, i);
    else a.hi = dfl_writelane(a.hi, v, i - 64u);
}

__device__ __forceinline__ void la_pqdownheap(DflLA &heap, const DflLA &key, uint32_t heap_len, uint32_t k)
{
    const uint32_t v = la_get(heap, k), kv = la_get(key, v);
    uint32_t j = k << 1;
    while (j <= heap_len) {
        uint32_t hj = la_get(heap, j), kj = la_get(key, hj);
        if (j < heap_len) {
            const uint32_t h2 = la_get(heap, j + 1u), k2 = la_get(key, h2);
            if (k2 <= kj) { j++; hj = h2; kj = k2; }
        }
        if (kv <= kj) break;
        la_set(heap, k, hj);
        k = j;
        j <<= 1;
    }
    la_set(heap, k, v);
}

// kind 0 / 1 / 2 as dfl_build_tree.  In: m natural leaves (code[k] ascending in CODE.lo lanes 0..m-1, key[k] =
// freq << 8), max_code of them (-1 if none).  Out: LEN[slot] for all leaves, the leaf count incl. forced ones in m,
// returns zlib's max_code.  Adds to opt_len / static_len exactly as build_tree + gen_bitlen do.
template <int KIND>
__device__ __forceinline__ int dfl_tree_small(uint32_t &m, int max_code, DflLA &CODE, DflLA &KEY, DflLA &LEN,
                                              long &opt_len, long &static_len)
{
    const uint32_t max_length = KIND == 2 ? 7u : 15u;
    while (m < 2u) {                                          // zlib forces two codes of non-zero frequency
        const uint32_t code = max_code < 2 ? (uint32_t)++max_code : 0u;
        la_set(CODE, m, code);
        la_set(KEY, m, 1u << 8);
        m++;
        opt_len--;
        if (KIND == 0) static_len -= (long)dfl_static_llen(code);
        else if (KIND == 1) static_len -= 5;
    }
    DflLA HEAP{0u, 0u}, DAD{0u, 0u};
    uint32_t heap_len = m;
    // heap[k + 1] = k: lane arrays can be filled by all lanes at once
    {
        const uint32_t lane = (uint32_t)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        HEAP.lo = lane == 0u ? 0u : lane - 1u;                // entries 1..63 = 0..62
        HEAP.hi = lane + 63u;                                 // entries 64.. = 63..
    }
    for (uint32_t n = heap_len >> 1; n >= 1u; n--) la_pqdownheap(HEAP, KEY, heap_len, n);
    uint32_t node = m, heap_max = 2u * m;
    do {
        const uint32_t n = la_get(HEAP, 1u);
        la_set(HEAP, 1u, la_get(HEAP, heap_len));
        heap_len--;
        la_pqdownheap(HEAP, KEY, heap_len, 1u);
        const uint32_t mm = la_get(HEAP, 1u);
        la_set(HEAP, --heap_max, n);
        la_set(HEAP, --heap_max, mm);
        const uint32_t kn = la_get(KEY, n), km = la_get(KEY, mm);
        const uint32_t dn = kn & 255u, dm = km & 255u;
        la_set(KEY, node, (((kn >> 8) + (km >> 8)) << 8) | ((dn >= dm ? dn : dm) + 1u));
        la_set(DAD, n, node);
        la_set(DAD, mm, node);
        la_set(HEAP, 1u, node);
        node++;
        la_pqdownheap(HEAP, KEY, heap_len, 1u);
    } while (heap_len >= 2u);
    la_set(HEAP, --heap_max, la_get(HEAP, 1u));

    // gen_bitlen
    DflLA BLC{0u, 0u};                                        // bl_count[0..15] in lanes 0..15
    uint32_t overflow = 0, h;
    la_set(LEN, la_get(HEAP, heap_max), 0u);
    for (h = heap_max + 1u; h < 2u * m; h++) {
        const uint32_t n = la_get(HEAP, h);
        uint32_t bits = la_get(LEN, la_get(DAD, n)) + 1u;
        if (bits > max_length) { bits = max_length; overflow++; }
        la_set(LEN, n, bits);
        if (n >= m) continue;                                 // not a leaf
        la_set(BLC, bits, la_get(BLC, bits) + 1u);
        const uint32_t code = la_get(CODE, n), f = la_get(KEY, n) >> 8;
        uint32_t xbits = 0, slen = 0;
        if (KIND == 0) { if (code >= 257u) xbits = dfl_lextra(code - 257u); slen = dfl_static_llen(code); }
        else if (KIND == 1) { xbits = dfl_dextra(code); slen = 5u; }
        else xbits = code == 16u ? 2u : (code == 17u ? 3u : (code == 18u ? 7u : 0u));
        opt_len += (long)f * (long)(bits + xbits);
        if (KIND != 2) static_len += (long)f * (long)(slen + xbits);
    }
    if (overflow > 0u) {
        int ov = (int)overflow;
        do {
            uint32_t bits = max_length - 1u;
            while (la_get(BLC, bits) == 0u) bits--;
            la_set(BLC, bits, la_get(BLC, bits) - 1u);
            la_set(BLC, bits + 1u, la_get(BLC, bits + 1u) + 2u);
            la_set(BLC, max_length, la_get(BLC, max_length) - 1u);
            ov -= 2;
        } while (ov > 0);
        for (uint32_t bits = max_length; bits != 0u; bits--) {
            uint32_t n = la_get(BLC, bits);
            while (n != 0u) {
                const uint32_t mm = la_get(HEAP, --h);
                if (mm >= m) continue;
                const uint32_t old = la_get(LEN, mm);
                if (old != bits) {
                    opt_len += ((long)bits - (long)old) * (long)(la_get(KEY, mm) >> 8);
                    la_set(LEN, mm, bits);
                }
                n--;
            }
        }
    }
    return max_code;
}

// The used symbols of hist[base .. base + count) as leaves of a lane array (all lanes take part).
// Returns false when there are more than 64 of them.
__device__ __forceinline__ bool dfl_gather_leaves(const uint32_t *hist, uint32_t count, DflLA &CODE, DflLA &KEY, uint32_t &m, int &max_code)
{
    const uint32_t lane = (uint32_t)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    m = 0; max_code = -1;
    for (uint32_t c0 = 0; c0 < count; c0 += 64u) {
        const uint32_t idx = c0 + lane;
        const uint32_t f = idx < count ? hist[idx] : 0u;
        uint64_t mask = __builtin_amdgcn_ballot_w64(f != 0u);
        if (m + (uint32_t)__builtin_popcountll(mask) > 64u) return false;
        while (mask) {
            const uint32_t i = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            la_set(CODE, m, c0 + i);
            la_set(KEY, m, (uint32_t)__builtin_amdgcn_readlane((int)f, (int)i) << 8);
            max_code = (int)(c0 + i);
            m++;
        }
    }
    return true;
}

// opt_len / static_len of the block in L.hist with the lane-array trees; false if the alphabet is too big for them
// (the caller then uses dfl_block_lengths).  All lanes call it; the result is wave-uniform.
__device__ __forceinline__ bool dfl_block_lengths_small(const DflLds &L, uint32_t lane, long &opt_len, long &static_len)
{
    DflLA CODE{0u, 0u}, KEY{0u, 0u}, LEN{0u, 0u};
    uint32_t m; int maxc;
    if (!dfl_gather_leaves(L.hist, 286u, CODE, KEY, m, maxc)) return false;
    // the distance counts are needed after the code lengths have overwritten the head of the histogram: take them now
    const uint32_t dfreq = lane < 30u ? L.hist[DFL_DOFF + lane] : 0u;
    opt_len = 0; static_len = 0;
    const int lmax = dfl_tree_small<0>(m, maxc, CODE, KEY, LEN, opt_len, static_len);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < 288u; i += 64u) L.llen[i] = 0;
    if (lane < 32u) L.dlen[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < m) L.llen[CODE.lo] = (uint8_t)LEN.lo;          // leaf slot = lane (m <= 64)
    // distance tree
    {
        DflLA DC{0u, 0u}, DK{0u, 0u}, DL{0u, 0u};
        uint32_t dm = 0; int dmaxc = -1;
        uint64_t mask = __builtin_amdgcn_ballot_w64(dfreq != 0u);
        while (mask) {
            const uint32_t i = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            la_set(DC, dm, i);
            la_set(DK, dm, (uint32_t)__builtin_amdgcn_readlane((int)dfreq, (int)i) << 8);
            dmaxc = (int)i;
            dm++;
        }
        const int dmax = dfl_tree_small<1>(dm, dmaxc, DC, DK, DL, opt_len, static_len);
        if (lane < dm) L.dlen[DC.lo] = (uint8_t)DL.lo;
        __builtin_amdgcn_wave_barrier();
        // run-length statistics of the two length arrays (zlib's scan_tree), by lane 0, into L.freq[0..19)
        if (lane == 0) {
            for (int n = 0; n < 19; n++) L.blf[n] = 0;
            dfl_scan_tree(L.llen, lmax, L.blf);
            dfl_scan_tree(L.dlen, dmax, L.blf);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // bit-length tree
    {
        DflLA BC{0u, 0u}, BK{0u, 0u}, BL{0u, 0u};
        const uint32_t bf = lane < 19u ? (uint32_t)L.blf[lane] : 0u;
        uint32_t bm = 0; int bmaxc = -1;
        uint64_t mask = __builtin_amdgcn_ballot_w64(bf != 0u);
        while (mask) {
            const uint32_t i = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            la_set(BC, bm, i);
            la_set(BK, bm, (uint32_t)__builtin_amdgcn_readlane((int)bf, (int)i) << 8);
            bmaxc = (int)i;
            bm++;
        }
        dfl_tree_small<2>(bm, bmaxc, BC, BK, BL, opt_len, static_len);
        // which of the 19 codes got a length: bit `code` of used
        const uint64_t used = __builtin_amdgcn_ballot_w64(lane < bm && BL.lo != 0u);
        uint32_t usedcodes = 0;
        {
            uint64_t u = used;
            while (u) { const uint32_t i = (uint32_t)__builtin_ctzll(u); u &= u - 1ull; usedcodes |= 1u << la_get(BC, i); }
        }
        const uint8_t bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int max_blindex;
        for (max_blindex = 18; max_blindex >= 3; max_blindex--)
            if (usedcodes >> bl_order[max_blindex] & 1u) break;
        opt_len += 3 * ((long)max_blindex + 1) + 5 + 5 + 4;
    }
    return true;
}

// Start of zlib's window (stream position of window[0]) when the parser stands at loop top p0.
// The window slides by 32 KiB at the first loop top where fewer than 262 bytes of look-ahead are
// left in it; `n` enters because the last, partly filled window slides one byte earlier.
__device__ __forceinline__ uint32_t dfl_window_base(uint32_t p0, uint32_t n)
{
    uint32_t base = p0 >= 65275u ? ((p0 - 65275u) >> 15) << 15 : 0u;
    for (;;) {
        const uint32_t t = base + (n <= base + 65535u ? 65274u : 65275u);
        if (p0 < t) break;
        base += 32768u;
    }
    return base;
}

// ---------------------------------------------------------------------------------------------
// one parse job per wavefront
// ---------------------------------------------------------------------------------------------
#ifdef DFL_STAMP
__device__ unsigned long long dfl_stamp_buf[64 * 8];
#define DFL_T(v) v = clock64()
#else
#define DFL_T(v) do { } while (0)
#endif

struct DflWave {
    DflLds L;
    unsigned long long t_flush;
    uint32_t lane;
    // block accounting
    uint64_t bits;
    uint32_t bcount;          // symbols in the open block
    uint32_t block_start;     // stream position where it starts
    uint32_t nblk;            // blocks closed so far (stand-alone jobs store cumbits per block)
    uint32_t unsafe;
    // output of the stand-alone stream
    bool store;
    bool price;               // cut into blocks and price them
    bool keep_blocks;
    uint32_t *sym, *pos;
    uint64_t *cumbits;
    uint32_t nsym;
    uint32_t n;               // stream length
};

// Everything in DflWave but `lane` is wave-uniform.  What comes back from a call arrives in VGPRs; this tells the
// compiler again that the values are uniform, so that the parser state returns to SGPRs.
__device__ __forceinline__ uint32_t dfl_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t dfl_u64(uint64_t v) { return ((uint64_t)dfl_u32((uint32_t)(v >> 32)) << 32) | dfl_u32((uint32_t)v); }
template <typename P> __device__ __forceinline__ P *dfl_uptr(P *p) { return (P *)(uintptr_t)dfl_u64((uint64_t)(uintptr_t)p); }
__device__ __forceinline__ void dfl_hist_reset(DflWave &w)
{
    for (uint32_t i = w.lane; i < DFL_HIST; i += 64u) w.L.hist[i] = i == 256u ? 1u : 0u;
}

// Close the open block: p0 = loop top of the iteration that closes it, end = strstart at that moment.
__device__ __forceinline__ void dfl_flush_body(DflWave &w, bool last, uint32_t p0, uint32_t end)
{
    unsigned long long tf0 = 0, tf1 = 0; (void)tf0; (void)tf1;
    DFL_T(tf0);
    long opt_len = 0, static_len = 0;
    const bool small = dfl_block_lengths_small(w.L, w.lane, opt_len, static_len);
    if (w.lane == 0) {
        if (!small) {                                        // big alphabet: the workgroup's LDS tree arrays, one wave at a time
            while (atomicCAS(w.L.lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(8);
            dfl_block_lengths(w.L, opt_len, static_len);
            __threadfence_block();
            atomicExch(w.L.lock, 0u);
        }
        uint32_t opt_lenb = (uint32_t)((opt_len + 3 + 7) >> 3);
        const uint32_t static_lenb = (uint32_t)((static_len + 3 + 7) >> 3);
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        const uint32_t stored_len = end - w.block_start;
        const bool stored_wins = stored_len + 4u <= opt_lenb;
        const bool have_buf = w.block_start >= dfl_window_base(p0, w.n);
        uint32_t kind = (stored_wins && have_buf) ? 0u : (static_lenb == opt_lenb ? 1u : 2u);
        // would the decision change if the stream went on (pair streams reuse x's blocks)?
        const bool have_buf_inf = w.block_start >= dfl_window_base(p0, 0xFFFFFFFFu);
        w.L.misc[0] = kind;
        w.L.misc[1] = kind == 0u ? stored_len : (kind == 1u ? (uint32_t)static_len : (uint32_t)opt_len);
        w.L.misc[2] = (stored_wins && have_buf != have_buf_inf) ? 1u : 0u;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t kind = w.L.misc[0], val = w.L.misc[1];
    w.unsafe |= w.L.misc[2];
    if (kind == 0u) {
        w.bits += 3u;
        w.bits = (w.bits + 7ull) & ~7ull;
        w.bits += 32ull + 8ull * val;
    } else {
        w.bits += 3ull + val;
    }
    if (last) w.bits = (w.bits + 7ull) & ~7ull;
    if (w.keep_blocks && w.lane == 0) w.cumbits[w.nblk] = w.bits;
    w.nblk++;
    __builtin_amdgcn_wave_barrier();
    dfl_hist_reset(w);
    __builtin_amdgcn_wave_barrier();
    w.bcount = 0;
    w.block_start = end;
    DFL_T(tf1);
    w.t_flush += tf1 - tf0;
}

// The flush as a CALL (it is big, and rare: once per 16 383 symbols).  What it reads and changes goes in and out BY
// VALUE, in registers: a reference to the wave's state would pin that whole struct to scratch memory, and every
// symbol of the hot loop would then load and store its counters there (round 1's kernel did: 2.2 MB written per pair).
// The LDS pointers are derived again from the wave number instead of being passed.
struct DflBlockState {
    uint64_t bits; unsigned long long t_flush;
    uint64_t *cumbits;
    uint32_t bcount, block_start, nblk, unsafe, n, keep;
};
__device__ __forceinline__ void dfl_lds_pointers(DflLds &L, uint32_t wave)
{
    extern __shared__ __align__(16) uint8_t dfl_lds[];
    uint8_t *lds = dfl_lds + wave * L_WAVE;
    L.hist = (uint32_t *)(lds + L_HIST); L.llen = lds + L_LLEN; L.dlen = lds + L_DLEN;
    L.blf = (uint16_t *)(lds + L_BLF); L.misc = (uint32_t *)(lds + L_MISC);
    L.heap = (uint16_t *)(dfl_lds + G_HEAP); L.freq = (uint16_t *)(dfl_lds + G_FREQ); L.dad = (uint16_t *)(dfl_lds + G_DAD);
    L.len = (uint16_t *)(dfl_lds + G_LEN); L.depth = dfl_lds + G_DEPTH; L.lock = (uint32_t *)(dfl_lds + G_LOCK);
}
__device__ __attribute__((noinline)) DflBlockState dfl_flush_call(DflBlockState s, bool last, uint32_t p0, uint32_t end)
{
    DflWave w;
    dfl_lds_pointers(w.L, (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
    w.lane = threadIdx.x & 63u;
    w.bits = dfl_u64(s.bits); w.t_flush = dfl_u64(s.t_flush); w.cumbits = dfl_uptr(s.cumbits);
    w.bcount = dfl_u32(s.bcount); w.block_start = dfl_u32(s.block_start); w.nblk = dfl_u32(s.nblk); w.unsafe = dfl_u32(s.unsafe);
    w.n = dfl_u32(s.n); w.keep_blocks = dfl_u32(s.keep) != 0u;
    dfl_flush_body(w, last, p0, end);
    s.bits = w.bits; s.t_flush = w.t_flush; s.bcount = w.bcount; s.block_start = w.block_start; s.nblk = w.nblk; s.unsafe = w.unsafe;
    return s;
}
// The same call with the whole struct passed and returned by value.  Either form keeps the hot loop free of the
// per-symbol scratch traffic; which one the register allocator likes better differs between the two instantiations of
// the parse body (measured, pair-compressions/s at 1024 x 1 Mbp: gzip kernel 151.6 k wide / 130.1 k slim, zlib
// kernel 155.0 k wide / 167.2 k slim; again after the 16-byte index records: gzip 169.6 k wide / 141.0 k slim), so each uses
// its better one (-DDFL_WIDE_K=false builds the gzip kernel with the slim form).
__device__ __attribute__((noinline)) DflWave dfl_flush_call_wide(DflWave w, bool last, uint32_t p0, uint32_t end)
{
    dfl_flush_body(w, last, p0, end);
    return w;
}
template <bool WIDE>
__device__ __forceinline__ void dfl_flush(DflWave &w, bool last, uint32_t p0, uint32_t end)
{
    if (WIDE) {
        const DflWave r = dfl_flush_call_wide(w, last, p0, end);
        w.bits = dfl_u64(r.bits); w.t_flush = dfl_u64(r.t_flush);
        w.bcount = dfl_u32(r.bcount); w.block_start = dfl_u32(r.block_start); w.nblk = dfl_u32(r.nblk); w.unsafe = dfl_u32(r.unsafe);
        return;
    }
    DflBlockState s;
    s.bits = w.bits; s.t_flush = w.t_flush; s.cumbits = w.cumbits; s.bcount = w.bcount; s.block_start = w.block_start;
    s.nblk = w.nblk; s.unsafe = w.unsafe; s.n = w.n; s.keep = w.keep_blocks ? 1u : 0u;
    const DflBlockState r = dfl_flush_call(s, last, p0, end);
    // what comes back from a call arrives in VGPRs: tell the compiler again that it is wave-uniform
    w.bits = dfl_u64(r.bits); w.t_flush = dfl_u64(r.t_flush);
    w.bcount = dfl_u32(r.bcount); w.block_start = dfl_u32(r.block_start); w.nblk = dfl_u32(r.nblk); w.unsafe = dfl_u32(r.unsafe);
}

// One symbol from the parser (wave-uniform arguments).  is_match: len/dist valid.
template <bool WIDE>
__device__ __forceinline__ void dfl_emit(DflWave &w, bool is_match, uint32_t q, uint32_t lit, uint32_t len, uint32_t dist, bool tail = false)
{
    if (w.lane == 0) {
        if (w.price) {
            // LDS adds without a return value: nothing to wait for
            if (is_match) {
                atomicAdd(&w.L.hist[257u + dfl_lcode(len - 3u)], 1u);
                atomicAdd(&w.L.hist[DFL_DOFF + dfl_dcode(dist - 1u)], 1u);
            } else {
                atomicAdd(&w.L.hist[lit], 1u);
            }
        }
        if (w.store) {
            w.sym[w.nsym] = is_match ? (0x80000000u | ((len - 3u) << 16) | dist) : lit;
            w.pos[w.nsym] = q;
        }
    }
    w.nsym++;
    w.bcount++;
    // (the literal zlib tallies after its main loop never closes a block: the final flush does)
    if (w.price && w.bcount == DFL_BLOCK_SYMS && !tail) {
        __builtin_amdgcn_wave_barrier();
        dfl_flush<WIDE>(w, false, q + 1u, is_match ? q + len : q + 1u);
    }
}

// Add the symbols [a, b) of a stored stream to the wave's histogram: whole 128-symbol chunks as the difference of
// two checkpoints (each lane owns five histogram entries), the ragged ends symbol by symbol.
constexpr uint32_t DFL_CHK = 128u;
__device__ __forceinline__ void dfl_hist_scan(uint32_t *hist, const uint32_t *sym, uint32_t a, uint32_t b, uint32_t lane)
{
    for (uint32_t i = a + lane; i < b; i += 64u) {
        const uint32_t sv = sym[i];
        if (sv >> 31) {
            atomicAdd(&hist[257u + dfl_lcode((sv >> 16) & 0x7fffu)], 1u);
            atomicAdd(&hist[DFL_DOFF + dfl_dcode((sv & 0xffffu) - 1u)], 1u);
        } else {
            atomicAdd(&hist[sv], 1u);
        }
    }
}
__device__ __forceinline__ void dfl_hist_add_range(uint32_t *hist, const uint32_t *sym, const uint32_t *chk, uint32_t a, uint32_t b, uint32_t lane)
{
    const uint32_t ca = (a + DFL_CHK - 1u) / DFL_CHK, cb = b / DFL_CHK;
    if (ca >= cb) { dfl_hist_scan(hist, sym, a, b, lane); return; }
    dfl_hist_scan(hist, sym, a, ca * DFL_CHK, lane);
    dfl_hist_scan(hist, sym, cb * DFL_CHK, b, lane);
    const uint32_t *pa = chk + (size_t)ca * DFL_HIST, *pb = chk + (size_t)cb * DFL_HIST;
    for (uint32_t i = lane; i < DFL_HIST; i += 64u) {
        const uint32_t d = pb[i] - pa[i];
        if (d) atomicAdd(&hist[i], d);
    }
}

// Checkpoints of one sequence's stored stream.  One wave per sequence.
__global__ void __launch_bounds__(64) dfl_chk_kernel(DflTables T, uint32_t *chk_rw, uint32_t nseq)
{
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    if (g >= nseq) return;
    __shared__ uint32_t hist[DFL_HIST];
    const DflSeq q = T.seq[g];
    const uint32_t *sym = T.sym + q.soff;
    uint32_t *out = chk_rw + q.koff;
    for (uint32_t i = lane; i < DFL_HIST; i += 64u) hist[i] = 0u;
    __syncthreads();
    const uint32_t nchk = q.nsym / DFL_CHK + 1u;
    for (uint32_t c = 0; c < nchk; ++c) {
        for (uint32_t i = lane; i < DFL_HIST; i += 64u) out[(size_t)c * DFL_HIST + i] = hist[i];
        __syncthreads();
        const uint32_t a = c * DFL_CHK, b = a + DFL_CHK < q.nsym ? a + DFL_CHK : q.nsym;
        dfl_hist_scan(hist, sym, a, b, lane);
        __syncthreads();
    }
}

// USE_K: with the six-byte index in the match search (level 9);  SEG: the launch consists of segment jobs (mode 2)
template <bool USE_K, bool SEG>
__device__ __forceinline__ void dfl_parse_body(const DflTables &T, const DflJob *jobs, uint32_t njobs, uint32_t *out)
{
    extern __shared__ __align__(16) uint8_t dfl_lds[];
    // the wave index is uniform: telling the compiler so moves the whole parser state to SGPRs
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t jid = blockIdx.x * DFL_WAVES + wave;
    if (threadIdx.x == 0) *(uint32_t *)(dfl_lds + G_LOCK) = 0u;
    __syncthreads();                                         // (before any wave can leave)
    if (jid >= njobs) return;
    const DflJob job = jobs[jid];

    DflWave w;
    dfl_lds_pointers(w.L, wave);
    w.lane = lane;

    const DflSeq sx = T.seq[job.xi];
    const bool pair = job.yi >= 0;
    DflSeq sy = sx;
    if (pair) sy = T.seq[job.yi];
    DflStream S;
    S.X = T.bytes + sx.boff; S.lx = sx.len;
    S.Y = pair ? T.bytes + sy.boff : S.X; S.ly = pair ? sy.len : 0u;
    S.n = S.lx + S.ly;
    S.slid = S.n > 65536u;                                   // (updated at every loop top, see there)
    const uint32_t lx = S.lx, n = S.n;
    const uint32_t *occx = T.occ + sx.ioff, *bsx = T.bstart + (size_t)job.xi * (DFL_NHASH + 1u);
    const uint32_t *occy = T.occ + sy.ioff;
    const uint64_t *occ8x = T.occ8 + sx.ioff, *occ8y = T.occ8 + sy.ioff;
    const uint64_t *inv2x = T.inv2 + sx.ioff, *inv2y = T.inv2 + sy.ioff;
    const uint32_t *kbsx = T.k_bstart + (size_t)job.xi * 65537u;
    const DflKRec *krecx = T.k_rec + sx.ioff, *krecy = T.k_rec + sy.ioff;
    const uint64_t *kinv2x = T.k_inv2 + sx.ioff, *kinv2y = T.k_inv2 + sy.ioff;

    // the two seam positions whose 3-byte hash mixes x and y
    const bool has1 = pair && lx >= 1u && lx - 1u + 3u <= n;
    const bool has2 = pair && lx >= 2u && lx - 2u + 3u <= n;
    const uint32_t hs1 = has1 ? dfl_hash3(S, lx - 1u) : 0xFFFFFFFFu;
    const uint32_t hs2 = has2 ? dfl_hash3(S, lx - 2u) : 0xFFFFFFFFu;
    // the (up to) five seam positions whose six bytes mix x and y are in neither sequence's six-byte index:
    // a probe whose hash equals one of theirs takes the full chain walk instead
    // (lane i < 5 holds the hash of seam position lx - 1 - i: one VGPR and one ballot per probe instead of five uniform values)
    uint32_t ksl = 0xFFFFFFFFu;
    if (lane < 5u && pair && lx >= lane + 1u && lx - (lane + 1u) + 6u <= n) ksl = dfl_hash6(dfl_load8(S, lx - (lane + 1u)));

    w.n = n; w.unsafe = 0; w.nblk = 0; w.t_flush = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, ta = 0, tb = 0, acc_search = 0, acc_sync = 0, iters = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)ta; (void)tb; (void)acc_search; (void)acc_sync; (void)iters;
    DFL_T(t0);
    const uint32_t mode = job.mode;
    w.store = mode == 1u || mode == 2u;
    w.price = mode != 2u;
    w.sym = T.sym + sx.soff; w.pos = T.pos + sx.soff; w.cumbits = T.cumbits + sx.coff;
    if (mode == 2u) { w.sym = T.seg_sym + job.aux; w.pos = T.seg_pos + job.aux; }
    const bool keep_blocks = mode == 1u || mode == 3u;          // record the bit count after every block

    w.keep_blocks = keep_blocks;
    uint32_t p;
    const bool restart = mode == 0u && sx.unsafe == 0u && sx.rk != 0u && T.norestart == 0u;
    if (restart) {
        for (uint32_t i = lane; i < DFL_HIST; i += 64u) w.L.hist[i] = T.rhist[(size_t)job.xi * DFL_HIST + i];
        const uint32_t blocks_before = sx.rkb / DFL_BLOCK_SYMS;
        w.bits = blocks_before ? (T.cumbits + sx.coff)[blocks_before - 1u] : 0ull;
        w.nblk = blocks_before;
        w.bcount = sx.rk - sx.rkb;
        w.nsym = sx.rk;
        w.block_start = sx.rbpos;
        p = sx.rpos;
    } else {
        dfl_hist_reset(w);
        w.bits = 0; w.bcount = 0; w.nsym = 0; w.block_start = 0; p = 0;
        if (mode == 2u) p = job.p0;                              // a segment starts as if right behind a match
        if (mode == 3u) p = n;                                   // nothing to parse: the stream is stored
    }
    __builtin_amdgcn_wave_barrier();

    const bool try_sync = mode == 0u && pair && S.ly > 65536u && sy.nsym != 0u;
    const uint32_t *symy = T.sym + sy.soff, *posy = T.pos + sy.soff;
    uint32_t sync_k = mode == 3u ? 0u : 0xFFFFFFFFu;              // pricing = "synchronised" with the own stream at symbol 0
    const uint32_t seg_stop = (mode == 2u && job.p1 < n) ? job.p1 + DFL_SEG_SLACK : 0xFFFFFFFFu;
    // a segment stops behind the first match past seg_stop -- or, on data with hardly any match, when its
    // scratch stream is full (the stitch then finds no common point and the sequence is parsed serially)
    const uint32_t seg_cap = mode == 2u ? (job.p1 - job.p0) + DFL_SEG_SLACK + DFL_SEG_ROOM - 2u : 0xFFFFFFFFu;
    const uint32_t yoff = mode == 3u ? 0u : lx;                  // stream position of the streamed symbols' base
    const uint32_t ynsym = sy.nsym;

    uint32_t match_length = 2u, match_start = 0u;
    bool match_available = false;

    while (p < n) {
        if (SEG && w.nsym >= seg_cap) break;
        const uint32_t la = n - p;
        const uint32_t prev_length = match_length, prev_match = match_start;
        // bytes behind the end can be read from p + 258 > n on; by then a stream longer than the buffer has slid,
        // a shorter one slides at the first loop top >= 65 274 (65 275 for n = 65 536) -- dfl_window_base's rule
        S.slid = n > 65536u || p >= (n <= 65535u ? 65274u : 65275u);
        match_length = 2u;
        DFL_T(ta); iters++;
        if (la >= 3u && prev_length < T.lazy) {
            // ---- the chain of p: earlier positions with the same hash, most recent first ----
            const uint64_t s0 = dfl_load8(S, p);                 // la >= 3: its first three bytes are input
            const uint32_t h = ((((uint32_t)s0 & 0xffu) << 10) ^ ((((uint32_t)s0 >> 8) & 0xffu) << 5) ^
                                (((uint32_t)s0 >> 16) & 0xffu)) & 0x7fffu;
            uint32_t ny = 0, ybase = 0, nsp = 0, sp0 = 0, sp1 = 0, nx = 0, xtop = 0;
            if (p >= lx) {
                const uint64_t e = inv2y[p - lx];
                ny = (uint32_t)(e >> 32); ybase = (uint32_t)e - 1u;
                if (hs1 == h) { sp0 = lx - 1u; nsp = 1u; }
                if (hs2 == h) { if (nsp) sp1 = lx - 2u; else sp0 = lx - 2u; nsp++; }
                // x's bucket is needed only while the chain can still reach the seam
                if (pair && p - lx <= DFL_MAX_DIST) { nx = bsx[h + 1u] - bsx[h]; xtop = bsx[h + 1u] - 1u; }
            } else if (p + 3u <= lx) {
                const uint64_t e = inv2x[p];
                nx = (uint32_t)(e >> 32); xtop = (uint32_t)e - 1u;
            } else {
                if (p == lx - 1u && hs2 == h) { sp0 = lx - 2u; nsp = 1u; }
                nx = bsx[h + 1u] - bsx[h]; xtop = bsx[h + 1u] - 1u;
            }
            const uint32_t total = ny + nsp + nx;
            uint32_t chain = T.chain;
            if (prev_length >= T.good) chain >>= 2;
            const uint32_t nice = T.nice > la ? la : T.nice;
            uint32_t best_len = prev_length;
            bool searched = false, done = false;

            // ---- K-pass: only the chain members that share the probe's six-byte hash.  If it finds a match of
            // >= 6 bytes, that is what the full walk would return (the first longest member, or the first
            // that reaches nice_match, has >= 6 matching bytes, hence sits in this bucket; members are taken
            // in the same order, with their exact position j3 in the 3-byte chain for budget and head rule).
            bool kdone = false;
            const uint32_t match_start_in = match_start;
            if (USE_K && total != 0u && la >= DFL_MIN_LOOKAHEAD && (p >= lx || p + 6u <= lx)) {
                const uint32_t h6 = dfl_hash6(s0);
                const bool kspecial = pair && p + 5u >= lx && p <= lx + DFL_MAX_DIST + 5u &&
                                      __builtin_amdgcn_ballot_w64(ksl == h6) != 0ull;
                if (!kspecial) {
                    uint32_t kny = 0, kybase = 0, knx = 0, kxtop = 0;
                    if (p >= lx) {
                        const uint64_t e = kinv2y[p - lx];
                        kny = (uint32_t)(e >> 32); kybase = (uint32_t)e - 1u;
                        if (pair && p - lx <= DFL_MAX_DIST) { knx = kbsx[h6 + 1u] - kbsx[h6]; kxtop = kbsx[h6 + 1u] - 1u; }
                    } else {
                        const uint64_t e = kinv2x[p];
                        knx = (uint32_t)(e >> 32); kxtop = (uint32_t)e - 1u;
                    }
                    const uint32_t ktotal = kny + knx;
                    const bool p_in_y = p >= lx;
                    // the bucket holds every earlier position of the sequence with this hash, the window only the
                    // most recent few (about 8 on DNA): look at 16 first, then 64 at a time
                    uint32_t width = DFL_KW;
                    for (uint32_t j0 = 0; j0 < ktotal; j0 += width, width = 64u) {
                        const uint32_t j = j0 + lane;
                        const bool in = j < ktotal && lane < width;
                        const bool fromy = j < kny;
                        uint32_t v = 0, r3 = 0;
                        uint64_t d8 = 0;
                        if (in) {
                            const uint32_t idx = fromy ? kybase - j : kxtop - (j - kny);
                            const DflKRec r = (fromy ? krecy : krecx)[idx];
                            v = r.v; d8 = r.d8; r3 = r.r3;
                        }
                        const uint32_t q = fromy ? lx + v : v;
                        // position of q in the 3-byte chain of p (0 = its head)
                        uint32_t j3;
                        if (p_in_y) j3 = fromy ? ny - 1u - r3 : ny + nsp + (nx - 1u - r3);
                        else j3 = nx - 1u - r3;
                        const uint32_t h3 = ((((uint32_t)d8 & 0xffu) << 10) ^ ((((uint32_t)d8 >> 8) & 0xffu) << 5) ^
                                             (((uint32_t)d8 >> 16) & 0xffu)) & 0x7fffu;
                        const bool near = in && p - q <= (j3 == 0u ? DFL_MAX_DIST : DFL_MAX_DIST - 1u);
                        const bool ok = near && q != 0u && h3 == h && j3 < chain;
                        const bool whole = v + 8u <= (fromy ? S.ly : lx);
                        uint64_t x8 = d8 ^ s0;
                        if (ok && !whole) x8 = dfl_load8(S, q) ^ s0;
                        const uint32_t len8 = (ok ? (x8 ? (uint32_t)__builtin_ctzll(x8) >> 3 : 8u) : 0u);
                        uint64_t after = ~0ull;
                        for (;;) {
                            const uint32_t floor8 = best_len < 7u ? best_len : 7u;
                            const uint64_t cand = __builtin_amdgcn_ballot_w64(len8 > floor8) & after;
                            if (!cand) break;
                            const uint32_t i = (uint32_t)__builtin_ctzll(cand);
                            after = i < 63u ? ~((2ull << i) - 1ull) : 0ull;
                            uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)len8, (int)i);
                            const uint32_t qi = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)i);
                            if (len == 8u) {
                                while (len < DFL_MAX_MATCH) {
                                    const uint64_t y8 = dfl_load8(S, p + len) ^ dfl_load8(S, qi + len);
                                    if (y8) { len += (uint32_t)__builtin_ctzll(y8) >> 3; break; }
                                    len += 8u;
                                }
                                if (len > DFL_MAX_MATCH) len = DFL_MAX_MATCH;
                            }
                            if (len > best_len) {
                                best_len = len;
                                match_start = qi;
                                if (len >= nice) { done = true; break; }
                            }
                        }
                        // the lists are in falling position order: behind the first member that is too far
                        // (or the last of the bucket) nothing can follow
                        if (done || __builtin_amdgcn_ballot_w64(in && !near) != 0ull) break;
                    }
                    if (best_len >= 6u && best_len > prev_length) {
                        kdone = true; searched = true;
                    } else if (prev_length >= 5u) {
                        // nothing can beat prev_length without >= 6 matching bytes: zlib emits the previous
                        // match whatever this search returns
                        kdone = true;
                    }
                    if (!kdone) { best_len = prev_length; done = false; match_start = match_start_in; }
                }
            }

            if (!kdone && total != 0u) {
                const uint32_t lim = total < chain ? total : chain;
                // 256 candidates per step, four per lane, their loads in flight together.  Every lane
                // gets the byte-exact common length within the first 8 bytes; the groups of 64 are then
                // walked in chain order by "first lane that beats the best so far" (one ballot per round,
                // each round raises the best length) -- zlib's own update rule.
                for (uint32_t j0 = 0; j0 < lim && !done; j0 += 256u) {
                    uint32_t q[4], len8[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j0 + 64u * (uint32_t)u >= lim) { ok[u] = false; len8[u] = 0u; q[u] = 0u; continue; }   // (uniform)
                        const uint32_t j = j0 + 64u * (uint32_t)u + lane;
                        const bool in = j < lim;
                        const bool fromy = j < ny;
                        const bool special = !fromy && j - ny < nsp;
                        uint32_t v = 0;
                        uint64_t d8 = 0;
                        if (in && !special) {                    // position and its 8 bytes: two coalesced reads
                            const uint32_t idx = fromy ? ybase - j : xtop - (j - ny - nsp);
                            v = (fromy ? occy : occx)[idx];
                            d8 = (fromy ? occ8y : occ8x)[idx];
                        }
                        q[u] = fromy ? lx + v : (special ? (j == ny ? sp0 : sp1) : v);
                        ok[u] = in && q[u] != 0u && p - q[u] <= (j == 0u ? DFL_MAX_DIST : DFL_MAX_DIST - 1u);
                        // the stored bytes are the stream's only if they do not run over their sequence's end
                        const bool whole = !special && v + 8u <= (fromy ? S.ly : lx);
                        uint64_t x8 = d8 ^ s0;
                        if (ok[u] && !whole) x8 = dfl_load8(S, q[u]) ^ s0;
                        len8[u] = x8 ? (uint32_t)__builtin_ctzll(x8) >> 3 : 8u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (done || j0 + 64u * (uint32_t)u >= lim) break;
                        // the walk ends at the first candidate that is out of range
                        const uint64_t bad = __builtin_amdgcn_ballot_w64(!ok[u]);
                        const uint32_t nvalid = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
                        if (!searched && nvalid == 0u) { done = true; break; }   // hash_head unusable: no search at all
                        searched = true;
                        const uint64_t live = nvalid < 64u ? (1ull << nvalid) - 1ull : ~0ull;
                        uint64_t after = ~0ull;                   // lanes behind the last one looked at
                        for (;;) {
                            const uint32_t floor8 = best_len < 7u ? best_len : 7u;
                            const uint64_t cand = __builtin_amdgcn_ballot_w64(len8[u] > floor8) & live & after;
                            if (!cand) break;
                            const uint32_t i = (uint32_t)__builtin_ctzll(cand);
                            after = i < 63u ? ~((2ull << i) - 1ull) : 0ull;
                            uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)len8[u], (int)i);
                            const uint32_t qi = (uint32_t)__builtin_amdgcn_readlane((int)q[u], (int)i);
                            if (len == 8u) {                      // 8 equal bytes: keep comparing (uniform addresses)
                                while (len < DFL_MAX_MATCH) {
                                    const uint64_t y8 = dfl_load8(S, p + len) ^ dfl_load8(S, qi + len);
                                    if (y8) { len += (uint32_t)__builtin_ctzll(y8) >> 3; break; }
                                    len += 8u;
                                }
                                if (len > DFL_MAX_MATCH) len = DFL_MAX_MATCH;
                            }
                            if (len > best_len) {
                                best_len = len;
                                match_start = qi;
                                if (len >= nice) { done = true; break; }
                            }
                        }
                        if (nvalid < 64u) done = true;
                    }
                }
            }
            if (searched) {
                match_length = best_len <= la ? best_len : la;
                if (match_length == 3u && p - match_start > DFL_TOO_FAR) match_length = 2u;
            }
        }
        DFL_T(tb); acc_search += tb - ta;
        if (prev_length >= 3u && match_length <= prev_length) {
            const uint32_t q = p - 1u;
            dfl_emit<DFL_WIDE_K && USE_K>(w, true, q, 0u, prev_length, q - prev_match);
            p = q + prev_length;
            match_available = false;
            match_length = 2u;
            if (p >= seg_stop) break;                             // segment: far enough into the next one
            // both parsers right behind a match, and x out of reach: from here on y's own stream
            if (try_sync && p >= lx + DFL_MAX_DIST + 1u && p < n) {
                const uint32_t want = p - lx;
                DFL_T(ta);
                uint32_t lo = 0, hi = sy.nsym;                   // first k with posy[k] >= want
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (posy[mid] < want) lo = mid + 1u; else hi = mid; }
                DFL_T(tb); acc_sync += tb - ta;
                if (lo < sy.nsym && lo > 0u && posy[lo] == want && (symy[lo - 1u] >> 31)) { sync_k = lo; break; }
            }
        } else if (match_available) {
            dfl_emit<DFL_WIDE_K && USE_K>(w, false, p - 1u, dfl_byte(S, p - 1u), 0u, 0u);
            p++;
        } else {
            match_available = true;
            p++;
        }
    }

    DFL_T(t1);
    const unsigned long long flush_parse = w.t_flush; (void)flush_parse;
    if (sync_k != 0xFFFFFFFFu) {
        // ---- y's own symbols from sync_k on, re-cut into this stream's blocks ----
        uint32_t k = sync_k;
        __builtin_amdgcn_wave_barrier();
        const uint32_t *chky = (mode == 0u && T.chk) ? T.chk + sy.koff : nullptr;
        while (k < ynsym) {
            uint32_t m = ynsym - k;
            if (!chky && m > 128u) m = 128u;                      // without checkpoints: two symbols per lane at a time
            if (m > DFL_BLOCK_SYMS - w.bcount) m = DFL_BLOCK_SYMS - w.bcount;
            if (chky) dfl_hist_add_range(w.L.hist, symy, chky, k, k + m, lane);     // everything up to the block's end at once
            else dfl_hist_scan(w.L.hist, symy, k, k + m, lane);
            k += m; w.bcount += m; w.nsym += m;
            if (w.bcount == DFL_BLOCK_SYMS) {
                const uint32_t sl = symy[k - 1u];
                const uint32_t q = yoff + posy[k - 1u];
                if (!(sl >> 31) && q + 1u == n) break;           // zlib's after-loop literal: see dfl_emit
                __builtin_amdgcn_wave_barrier();
                dfl_flush<DFL_WIDE_K && USE_K>(w, false, q + 1u, (sl >> 31) ? q + ((sl >> 16) & 0x7fffu) + 3u : q + 1u);
            }
        }
        __builtin_amdgcn_wave_barrier();
    } else if (match_available && p >= n) {
        dfl_emit<DFL_WIDE_K && USE_K>(w, false, p - 1u, dfl_byte(S, p - 1u), 0u, 0u, true);
    }
    __builtin_amdgcn_wave_barrier();
    DFL_T(t2);
    if (w.price) dfl_flush<DFL_WIDE_K && USE_K>(w, true, n, n);
    DFL_T(t3);
#ifdef DFL_STAMP
    if (lane == 0 && jid < 64u) {
        unsigned long long *d = dfl_stamp_buf + jid * 8u;
        d[0] = t1 - t0; d[1] = acc_search; d[2] = acc_sync; d[3] = flush_parse; d[4] = t2 - t1; d[5] = w.t_flush - flush_parse; d[6] = iters; d[7] = t3 - t0;
    }
#endif

    if (lane == 0) {
        // device-side check (read by dfl_check_status): a stored symbol stream must end inside its capacity --
        // len + 1 entries for a sequence's own stream, the segment's share of the scratch for a segment job
        if (w.store && w.nsym > (mode == 2u ? (job.p1 - job.p0) + DFL_SEG_SLACK + DFL_SEG_ROOM : S.lx + S.ly + 1u))
            atomicOr(T.status, DFL_ST_CAP);
        if (mode == 2u) {
            T.seg_cnt[job.out_idx] = w.nsym;
        } else {
            out[job.out_idx] = (uint32_t)(w.bits >> 3);
            if (mode != 0u) {
                DflSeq *d = T.seq + job.xi;
                d->nsym = w.nsym;
                d->unsafe = w.unsafe;
                d->total_bits = w.bits;
            }
        }
    }
}

// The two kernels: with the six-byte index (gzip) the body needs a few more registers and runs at 7 waves per SIMD;
// without it (zlib) at 8.  Measured, pair-compr/s at 1024 x 1 Mbp: gzip 4 / 5 / 6 / 7 waves 138 / 167 / 167 / 170 k --
// beyond 5 waves the kernel is bound by the memory system (distinct lines per probe), not by latency; zlib 4 / 5 / 8
// waves 139 / 139 / 167 k.  The K-pass looks at 16 bucket members first (8: 153 k, 12: 164 k, 16: 167 k at 5 waves).
#ifndef DFL_WPE_K
#define DFL_WPE_K 7
#endif
#ifndef DFL_WPE
#define DFL_WPE 8
#endif
template <bool SEG>
__global__ void __launch_bounds__(64 * DFL_WAVES) __attribute__((amdgpu_waves_per_eu(DFL_WPE_K)))
dfl_parse_kernel_k(DflTables T, const DflJob *jobs, uint32_t njobs, uint32_t *out)
{
    dfl_parse_body<true, SEG>(T, jobs, njobs, out);
}
template <bool SEG>
__global__ void __launch_bounds__(64 * DFL_WAVES) __attribute__((amdgpu_waves_per_eu(DFL_WPE)))
dfl_parse_kernel(DflTables T, const DflJob *jobs, uint32_t njobs, uint32_t *out)
{
    dfl_parse_body<false, SEG>(T, jobs, njobs, out);
}

// Job list of a row tile, made on the device: job t = pair (r0 + t / n, t % n), result slot t.
__global__ void dfl_rowjobs_kernel(DflJob *jobs, uint32_t r0, uint32_t nrows, uint32_t n)
{
    const uint64_t total = (uint64_t)nrows * n;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        DflJob j;
        j.xi = (int32_t)(r0 + (uint32_t)(t / n)); j.yi = (int32_t)(t % n);
        j.mode = 0u; j.out_idx = (uint32_t)t; j.p0 = 0u; j.p1 = 0u; j.aux = 0ull;
        jobs[t] = j;
    }
}

// Restart record of every sequence: the last clean state at least DFL_RESTART_BACK bytes before
// its end, and the symbol counts of the block that is open there.  One wave per sequence.
__global__ void __launch_bounds__(64) dfl_restart_kernel(DflTables T, uint32_t nseq)
{
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    if (g >= nseq) return;
    __shared__ uint32_t hist[DFL_HIST];
    DflSeq *d = T.seq + g;
    const uint32_t *sym = T.sym + d->soff, *pos = T.pos + d->soff;
    const uint32_t nsym = d->nsym, len = d->len;
    uint32_t k = 0;
    if (len > DFL_RESTART_BACK && nsym != 0u) {
        const uint32_t target = len - DFL_RESTART_BACK;
        uint32_t lo = 0, hi = nsym;                              // first k with pos[k] > target
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pos[mid] <= target) lo = mid + 1u; else hi = mid; }
        k = lo ? lo - 1u : 0u;
        while (k > 0u && !(sym[k - 1u] >> 31)) k--;
    }
    const uint32_t kb = k / DFL_BLOCK_SYMS * DFL_BLOCK_SYMS;
    for (uint32_t i = lane; i < DFL_HIST; i += 64u) hist[i] = i == 256u ? 1u : 0u;
    __syncthreads();
    for (uint32_t i = kb + lane; i < k; i += 64u) {
        const uint32_t s = sym[i];
        if (s >> 31) {
            atomicAdd(&hist[257u + dfl_lcode((s >> 16) & 0x7fffu)], 1u);
            atomicAdd(&hist[DFL_DOFF + dfl_dcode((s & 0xffffu) - 1u)], 1u);
        } else {
            atomicAdd(&hist[s], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = lane; i < DFL_HIST; i += 64u) T.rhist[(size_t)g * DFL_HIST + i] = hist[i];
    if (lane == 0) {
        d->rk = k;
        d->rpos = k < nsym ? pos[k] : 0u;
        d->rkb = kb;
        d->rbpos = kb < nsym ? pos[kb] : 0u;
        if (k >= nsym) d->rk = 0u;                               // nothing to restart from
    }
}

// Stitch of the segment streams of one sequence.  Segment t starts at p0[t] "as if right behind a
// match"; the stream of segment t-1 (which ran DFL_SEG_SLACK further) is the true one there.  From the
// first position at which BOTH stand right behind a match the two parsers are in the same state
// (state = position, the chains are data) and segment t's stream is the true continuation.
// One thread per segment; ends[t] = symbols of t-1 that are kept, from[t] = first kept symbol of t.
struct DflSeg { uint64_t aux; uint32_t p0, p1, first, cnt; };   // first: 1 = first segment of its sequence

__global__ void dfl_stitch_kernel(const DflSeg *seg, const uint32_t *cnt, uint32_t nseg, const uint32_t *ssym,
                                  const uint32_t *spos, uint32_t *from, uint32_t *ends, uint32_t *fail)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseg) return;
    if (seg[t].first) { from[t] = 0u; return; }
    const uint32_t *symA = ssym + seg[t - 1u].aux, *posA = spos + seg[t - 1u].aux;
    const uint32_t *symB = ssym + seg[t].aux, *posB = spos + seg[t].aux;
    const uint32_t cntA = cnt[t - 1u], cntB = cnt[t], s0 = seg[t].p0;
    uint32_t lo = 0, hi = cntA;                               // first a with posA[a] >= s0
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (posA[mid] < s0) lo = mid + 1u; else hi = mid; }
    uint32_t a = lo, b = 0;
    // where A's stream ends (it stops right behind a match unless it ran to the end of the sequence)
    uint32_t endA = 0;
    if (cntA) { const uint32_t sl = symA[cntA - 1u]; endA = posA[cntA - 1u] + ((sl >> 31) ? ((sl >> 16) & 0x7fffu) + 3u : 1u); }
    bool found = false;
    while (b < cntB) {
        const uint32_t pb = posB[b];
        const bool cleanB = b == 0u || (symB[b - 1u] >> 31);
        while (a < cntA && posA[a] < pb) a++;
        if (a < cntA) {
            if (posA[a] == pb && cleanB && a > 0u && (symA[a - 1u] >> 31)) { found = true; break; }
        } else {
            if (pb == endA && cleanB && cntA && (symA[cntA - 1u] >> 31)) { found = true; break; }
            if (pb > endA) break;
        }
        b++;
    }
    if (found) { ends[t - 1u] = a; from[t] = b; }
    else { ends[t - 1u] = 0u; from[t] = 0u; atomicOr(fail + t, 1u); }
}

__global__ void dfl_compact_kernel(const DflSeg *seg, const uint32_t *from, const uint32_t *num, const uint64_t *dst,
                                   const uint32_t *ssym, const uint32_t *spos, uint32_t *sym, uint32_t *pos)
{
    const uint32_t t = blockIdx.x;
    const uint32_t *a = ssym + seg[t].aux + from[t], *b = spos + seg[t].aux + from[t];
    uint32_t *da = sym + dst[t], *db = pos + dst[t];
    for (uint32_t i = threadIdx.x; i < num[t]; i += blockDim.x) { da[i] = a[i]; db[i] = b[i]; }
}

}  // namespace
