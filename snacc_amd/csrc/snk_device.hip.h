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
    const uint8_t  *const *bytes;     // ASCII, padded
    const uint8_t  *const *packed;    // 2-bit, padded (NULL entries for non-ACGT sequences)
    const uint32_t *len;
    const uint32_t *snap_pos;         // block-aligned prefix length covered by the snapshot (0 = none)
    uint32_t       *snap_out;         // frame bytes (header included) after snap_pos
    uint32_t       *snap_fast;        // [n][1024] 5-mer indexed tables (ACGT sequences)
    uint32_t       *snap_gen;         // [n][4096] hash indexed tables
    const uint32_t *lut_partner;      // [1024]  3 x 10-bit colliding 5-mer codes (self when none)
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

struct SnkFastSrc {
    const uint8_t *xp, *yp;   // packed; base i of a sequence at bits 2(i&3) of byte i>>2
    uint32_t lx;
};

// 64-bit window over the virtual concatenation x+y: bases [p-8, p+21) with
// base p-8 at bit 0 (29 bases valid, top 6 bits zero).
__device__ __forceinline__ uint64_t snk_fetchw_straddle(const SnkFastSrc &s, uint32_t p)
{
    int32_t q = (int32_t)p - 8;                         // q < lx < q + 29
    uint64_t xv = snk_ld8u(s.xp + (q >> 2)) >> ((q & 3) * 2);   // zero beyond lx (padding)
    uint32_t sh = 2u * (uint32_t)((int32_t)s.lx - q);   // 2..56
    uint64_t yv = snk_ld8u(s.yp) << sh;
    return (xv | yv) & 0x03FFFFFFFFFFFFFFull;
}

__device__ __forceinline__ uint64_t snk_fetchw(const SnkFastSrc &s, uint32_t p)
{
    int32_t q0 = (int32_t)p - 8;
    bool inx = (p + 21u <= s.lx);
    bool iny = (q0 >= (int32_t)s.lx);
    if (__builtin_expect(inx | iny, 1)) {
        int32_t q = iny ? q0 - (int32_t)s.lx : q0;
        const uint8_t *b = iny ? s.yp : s.xp;
        return snk_ld8u(b + (q >> 2)) >> ((q & 3) * 2);
    }
    return snk_fetchw_straddle(s, p);
}

__device__ __forceinline__ uint32_t snk_base_at(const SnkFastSrc &s, uint32_t p)
{
    return (uint32_t)snk_fetchw(s, p + 8u) & 3u;
}

// put(): table[k] = pos, plus the 5-mers that share k's slot in liblz4's hash
__device__ __forceinline__ void snk_fast_put(uint32_t *tbl, uint32_t k, uint32_t pk, uint32_t pos)
{
    tbl[k] = pos;
    uint32_t p1 = pk & 1023u, p2 = (pk >> 10) & 1023u, p3 = (pk >> 20) & 1023u;
    if (p1 != k) {
        tbl[p1] = pos;
        if (p2 != k) {
            tbl[p2] = pos;
            if (p3 != k) tbl[p3] = pos;
        }
    }
}

// One chain of the 2-bit kernel.  `tbl` = this chain's 1024 x u32 LDS table,
// `lut` = the workgroup's partner LUT in LDS.
__device__ __forceinline__ void snk_fast_chain(const SnkTables &T, const SnkJob job,
                                               uint32_t *tbl, const uint32_t *lut,
                                               uint32_t *out, uint32_t *status)
{
    SnkFastSrc s;
    const uint32_t lx = T.len[job.xi];
    const uint32_t ly = job.yi >= 0 ? T.len[job.yi] : 0u;
    const uint32_t n = lx + ly;
    s.xp = T.packed[job.xi];
    s.yp = job.yi >= 0 ? T.packed[job.yi] : T.zero_pad + SNK_PAD;
    s.lx = lx;

    uint32_t pos, total;
    const uint32_t spos = T.snap_pos[job.xi];
    if (job.snap == 0 && spos != 0u) { pos = spos; total = T.snap_out[job.xi]; }
    else                             { pos = 0u;   total = T.header_bytes; }

    // per-block state
    uint32_t cur = 0, step = 1, nb = 64, anchor = 0, op = 0;
    uint32_t iend = 0, mfl1 = 0, mlimit = 0, olimit = 0, blen = 0;
    bool pending = false;       // put(ip-2) owed before the next probe
    bool in_block = false;
    uint64_t guard = 2ull * n + 4096ull;

    for (;;) {
        if (!in_block) {
            // ---------------- block transition (once per 64 KiB) ----------------
            if (job.snap != 0 && pos == spos && spos != 0u) {
                uint32_t *dst = T.snap_fast + (size_t)job.xi * 1024u;
                for (uint32_t t = 0; t < 1024u; ++t) dst[t] = tbl[t];
                T.snap_out[job.xi] = total;
            }
            if (pos >= n) break;
            blen = n - pos < SNK_BLOCK ? n - pos : SNK_BLOCK;
            iend = pos + blen;
            if (blen < 13u) {                       // always stored raw; table untouched
                total += 4u + blen;
                pos = iend;
                continue;
            }
            mfl1 = iend - 11u; mlimit = iend - 5u; olimit = blen - 1u;
            {
                uint64_t w = snk_fetchw(s, pos);
                uint32_t k = (uint32_t)(w >> 16) & 1023u;
                snk_fast_put(tbl, k, lut[k], pos);
            }
            cur = pos + 1u; step = 1u; nb = 64u; anchor = pos; op = 0u;
            pending = false; in_block = true;
        }
        if (--guard == 0) { atomicOr(status, SNK_ST_ITERCAP); break; }

        // ---------------- one probe ----------------
        const uint32_t next = cur + step;
        bool bail = false, last = false;
        if (next > mfl1) {
            last = true;
        } else {
            const uint64_t wc = snk_fetchw(s, cur);
            if (pending) {
                uint32_t k2 = (uint32_t)(wc >> 12) & 1023u;       // 5-mer at cur-2
                snk_fast_put(tbl, k2, lut[k2], cur - 2u);
            }
            const uint32_t k = (uint32_t)(wc >> 16) & 1023u;       // 5-mer at cur
            uint32_t cand = tbl[k];
            snk_fast_put(tbl, k, lut[k], cur);
            { uint32_t s2 = nb >> 6; step = s2 ? s2 : 1u; nb++; }

            const uint64_t wd = snk_fetchw(s, cand);
            const uint64_t x = wc ^ wd;
            const uint64_t fw = (x >> 16) | (1ull << 42);          // 21 forward bases
            uint32_t f = (uint32_t)__builtin_ctzll(fw) >> 1;       // equal bases from cur
            const bool near = cand + SNK_MAXDIST >= cur;
            if (near && f >= 4u) {
                // ---------------- match ----------------
                uint32_t ip = cur;
                uint32_t lit = ip - anchor;
                if (lit != 0u && cand != 0u) {                     // catch up
                    uint32_t t = (uint32_t)x & 0xFFFFu;
                    uint32_t eq = t ? ((uint32_t)__builtin_clz(t << 16) >> 1) : 8u;
                    uint32_t b = eq < lit ? eq : lit;
                    b = b < cand ? b : cand;
                    ip -= b; cand -= b; lit -= b;
                    if (b == 8u) {
                        while (ip > anchor && cand > 0u &&
                               snk_base_at(s, ip - 1u) == snk_base_at(s, cand - 1u)) { ip--; cand--; lit--; }
                    }
                }
                op += 1u;
                if (op + lit + 8u + lit / 255u > olimit) bail = true;
                else {
                    op += lit + snk_lit_ext(lit) + 2u;
                    uint32_t e = cur + f;
                    if (f == 21u) {                                // long match: keep counting
                        uint32_t bpos = cand + (cur - ip) + 21u;
                        while (e < mlimit) {
                            uint64_t d = snk_fetchw(s, e + 8u) ^ snk_fetchw(s, bpos + 8u);
                            uint32_t c = d ? ((uint32_t)__builtin_ctzll(d) >> 1) : 29u;
                            if (c > 29u) c = 29u;
                            e += c; bpos += c;
                            if (c < 29u) break;
                        }
                    }
                    if (e > mlimit) e = mlimit;
                    const uint32_t mc = e - (ip + 4u);
                    if (op + 6u + (mc + 240u) / 255u > olimit) bail = true;
                    else {
                        if (mc >= 15u) op += (mc - 15u) / 255u + 1u;
                        anchor = e;
                        cur = e; step = 1u; nb = 63u; pending = true;
                        if (e >= mfl1) last = true;
                    }
                }
            } else {
                cur = next; pending = false;
            }
        }
        if (bail | last) {
            uint32_t payload = blen;
            if (!bail) {
                uint32_t run = iend - anchor;
                if (op + run + 1u + (run + 240u) / 255u <= olimit)
                    payload = op + 1u + snk_lit_ext(run) + run;
            }
            total += 4u + payload;
            pos = iend;
            in_block = false;
        }
    }
    out[job.out_idx] = total + 4u;      // end mark
}

// grid: one workgroup per `lanes*waves` jobs.  dynamic LDS = 4 KiB LUT + 4 KiB per chain.
__global__ void snk_fast_kernel(SnkTables T, const SnkJob *jobs, uint32_t n_jobs,
                                uint32_t lanes, uint32_t *out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t snk_lds[];
    uint32_t *lut = snk_lds;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t chains = lanes * waves;

    for (uint32_t t = tid; t < 1024u; t += blockDim.x) lut[t] = T.lut_partner[t];

    // chain c of the workgroup -> lane c / waves of wave c % waves  (spreads a
    // partially filled tail group over all waves)
    const uint32_t c = lane * waves + wave;
    const uint32_t j = blockIdx.x * chains + c;
    const bool active = lane < lanes && j < n_jobs;
    uint32_t *tbl = snk_lds + 1024u + (size_t)(wave * lanes + (lane < lanes ? lane : 0u)) * 1024u;

    SnkJob job; job.xi = 0; job.yi = -1; job.out_idx = 0; job.snap = 0;
    if (active) job = jobs[j];

    // cooperative table initialisation: snapshot of the prefix sequence, or zeros
    for (uint32_t l = 0; l < lanes; ++l) {
        const int a   = __shfl((int)active, (int)l);
        const int xi  = __shfl(job.xi, (int)l);
        const int snp = __shfl(job.snap, (int)l);
        if (!a) continue;
        uint32_t *dst = snk_lds + 1024u + (size_t)(wave * lanes + l) * 1024u;
        const bool use = (snp == 0) && (T.snap_pos[xi] != 0u);
        const uint32_t *src = T.snap_fast + (size_t)xi * 1024u;
        for (uint32_t t = lane; t < 1024u; t += 64u) dst[t] = use ? src[t] : 0u;
    }
    __syncthreads();

    if (active) snk_fast_chain(T, job, tbl, lut, out, status);
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

// 5-mer-indexed snapshot -> hash-indexed snapshot (for ACGT prefix + non-ACGT suffix pairs)
__global__ void snk_snap_convert_kernel(const uint32_t *snap_fast, uint32_t *snap_gen,
                                        const uint32_t *lut_hash, const uint32_t *seq_ids, uint32_t n_ids)
{
    const uint32_t g = seq_ids[blockIdx.x];
    (void)n_ids;
    uint32_t *dst = snap_gen + (size_t)g * 4096u;
    const uint32_t *src = snap_fast + (size_t)g * 1024u;
    for (uint32_t t = threadIdx.x; t < 4096u; t += blockDim.x) dst[t] = 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 1024u; k += blockDim.x) dst[lut_hash[k]] = src[k];
}
