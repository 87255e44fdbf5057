// snk_legacy.hip.h -- legacy nested-loop byte kernel (snk_generic_kernel): u32/u16 tables as in liblz4; an independent second implementation used for cross-checks (bytes_legacy=1).
// Part of the device code of libsnacc_hip.so; see snk_common.hip.h for the execution model.
#pragma once
#include "snk_common.hip.h"

// =========================================================================
//  generic byte kernel
// =========================================================================

// (SnkGenSrc, snk_byte_at, snk_ld8, snk_hash5, snk_hash4: snk_common.hip.h)

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

    if (n == 0u) { out[job.out_idx] = 7u + 4u; return; }        // liblz4 drops the content-size field of an empty frame
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
