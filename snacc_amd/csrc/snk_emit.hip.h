// snk_emit.hip.h -- LZ4 *frame bytes* (not only sizes) on the GPU, for -s/--save-compression with
// lz4 (ref:snacc/pairwise_ncd.py:82-88 writes the blob lz4framed.compress returned).
// Part of the device code of libsnacc_hip.so; see snk_common.hip.h for the execution model.
//
// Plain nested-loop encoder (the shape of liblz4's own loop) built on the legacy kernel's helpers:
// speed is irrelevant here (the reference writes N*N files), exactness is not -- the bytes equal
// liblz4 1.9.3 LZ4F_compressFrame(prefs=NULL) (tests compare with the binary and decode them back).
#pragma once
#include "snk_legacy.hip.h"

struct SnkEmitJob {
    int32_t  xi, yi;          // yi = -1: single sequence
    uint64_t off;             // where the frame starts in the output buffer
};

__device__ __forceinline__ uint8_t *snk_emit_len(uint8_t *op, uint32_t len)   // len already reduced by 15
{
    for (; len >= 255u; len -= 255u) *op++ = 255u;
    *op++ = (uint8_t)len;
    return op;
}

// One block; writes the compressed bytes to `dst` and returns their count, or returns blen when
// liblz4's limitedOutput compressor gives up (caller stores the block raw).
template <bool LINKED>
__device__ __forceinline__ uint32_t snk_emit_block(const SnkGenSrc &s, uint32_t *tbl, uint32_t pos, uint32_t blen,
                                                   uint8_t *dst, uint64_t &guard, uint32_t *status)
{
    const uint32_t iend = pos + blen;
    if (blen < 13u) return blen;
    const uint32_t mfl1 = iend - 11u, mlimit = iend - 5u, olimit = blen - 1u;
    uint32_t cur, step = 1u, nb = 64u, anchor = pos;
    uint8_t *op = dst;
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
            uint8_t *token = op++;
            if ((uint32_t)(op - dst) + lit + 8u + lit / 255u > olimit) return blen;
            if (lit >= 15u) { *token = 0xF0u; op = snk_emit_len(op, lit - 15u); }
            else            { *token = (uint8_t)(lit << 4); }
            for (uint32_t k = 0; k < lit; ++k) *op++ = (uint8_t)snk_byte_at(s, anchor + k);
            const uint32_t dist = ip - cand;
            *op++ = (uint8_t)dist; *op++ = (uint8_t)(dist >> 8);
            uint32_t a = ip + 4u, b = cand + 4u;
            while (a < mlimit) {
                uint64_t d = snk_ld8(s, a) ^ snk_ld8(s, b);
                if (d) { a += (uint32_t)__builtin_ctzll(d) >> 3; break; }
                a += 8u; b += 8u;
            }
            if (a > mlimit) a = mlimit;
            const uint32_t mc = a - (ip + 4u);
            if ((uint32_t)(op - dst) + 6u + (mc + 240u) / 255u > olimit) return blen;
            if (mc >= 15u) { *token |= 15u; op = snk_emit_len(op, mc - 15u); }
            else           { *token |= (uint8_t)mc; }
            anchor = a;
            cur = a; step = 1u; nb = 63u; pending = true;
            if (a >= mfl1) break;
        } else {
            cur = next; pending = false;
        }
    }
    {
        const uint32_t run = iend - anchor;
        if ((uint32_t)(op - dst) + run + 1u + (run + 240u) / 255u > olimit) return blen;
        if (run >= 15u) { *op++ = 0xF0u; op = snk_emit_len(op, run - 15u); }
        else            { *op++ = (uint8_t)(run << 4); }
        for (uint32_t k = 0; k < run; ++k) *op++ = (uint8_t)snk_byte_at(s, anchor + k);
    }
    return (uint32_t)(op - dst);
}

// XXH32 (seed 0) of fewer than 16 bytes: the frame descriptor checksum of the LZ4 frame format.
__device__ __forceinline__ uint32_t snk_xxh32_short(const uint8_t *p, uint32_t len)
{
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    uint32_t h = P5 + len, i = 0;
    for (; i + 4u <= len; i += 4u) {
        const uint32_t w = (uint32_t)p[i] | ((uint32_t)p[i + 1] << 8) | ((uint32_t)p[i + 2] << 16) | ((uint32_t)p[i + 3] << 24);
        h += w * P3; h = ((h << 17) | (h >> 15)) * P4;
    }
    for (; i < len; ++i) { h += p[i] * P5; h = ((h << 11) | (h >> 21)) * P1; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ void snk_emit_chain(const SnkTables &T, const SnkEmitJob job, uint32_t *tbl,
                                               uint8_t *frames, uint32_t *sizes, uint32_t idx, uint32_t *status)
{
    SnkGenSrc s;
    const uint32_t lx = T.len[job.xi];
    const uint32_t ly = job.yi >= 0 ? T.len[job.yi] : 0u;
    const uint32_t n = lx + ly;
    s.xb = T.bytes[job.xi];
    s.yb = job.yi >= 0 ? T.bytes[job.yi] : T.zero_pad + SNK_PAD;
    s.lx = lx;
    uint64_t guard = 2ull * n + 4096ull;
    uint8_t *op = frames + job.off;
    const bool linked = n > SNK_BLOCK;
    // header: magic, FLG (version 01, block-independence bit for a single block, content-size bit),
    // BD (64 KiB), [content size, 8 bytes LE], HC = second byte of XXH32(FLG .. before HC, seed 0)
    op[0] = 0x04; op[1] = 0x22; op[2] = 0x4D; op[3] = 0x18;
    const bool csize = T.header_bytes == 15u && n != 0u;      // liblz4 drops the field when the content is empty
    op[4] = (uint8_t)((linked ? 0x40 : 0x60) | (csize ? 0x08 : 0x00)); op[5] = 0x40;
    uint32_t hl = 2u;
    if (csize) { for (uint32_t k = 0; k < 8u; ++k) op[6u + k] = k < 4u ? (uint8_t)(n >> (8u * k)) : 0u; hl = 10u; }
    op[4u + hl] = (uint8_t)(snk_xxh32_short(op + 4, hl) >> 8);
    op += 5u + hl;
    uint32_t pos = 0;
    while (pos < n) {
        const uint32_t blen = n - pos < SNK_BLOCK ? n - pos : SNK_BLOCK;
        uint32_t c = linked ? snk_emit_block<true>(s, tbl, pos, blen, op + 4, guard, status)
                            : snk_emit_block<false>(s, tbl, pos, blen, op + 4, guard, status);
        uint32_t word = c;
        if (c >= blen) {                                    // stored raw
            for (uint32_t k = 0; k < blen; ++k) op[4 + k] = (uint8_t)snk_byte_at(s, pos + k);
            c = blen; word = blen | 0x80000000u;
        }
        op[0] = (uint8_t)word; op[1] = (uint8_t)(word >> 8); op[2] = (uint8_t)(word >> 16); op[3] = (uint8_t)(word >> 24);
        op += 4u + c;
        pos += blen;
    }
    op[0] = op[1] = op[2] = op[3] = 0;                      // end mark
    op += 4;
    sizes[idx] = (uint32_t)(op - (frames + job.off));
}

// grid: one 64-thread workgroup per `chains` jobs; dynamic LDS = 16 KiB per chain (zeroed here).
__global__ void snk_emit_kernel(SnkTables T, const SnkEmitJob *jobs, uint32_t n_jobs, uint32_t chains,
                                uint8_t *frames, uint32_t *sizes, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t snk_lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t j = blockIdx.x * chains + lane;
    const bool active = lane < chains && j < n_jobs;
    for (uint32_t t = lane; t < chains * 4096u; t += 64u) snk_lds[t] = 0u;
    __syncthreads();
    if (active) snk_emit_chain(T, jobs[j], snk_lds + (size_t)lane * 4096u, frames, sizes, j, status);
}
