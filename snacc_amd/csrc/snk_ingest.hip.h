// snk_ingest.hip.h -- ingest helpers: classify, 2-bit pack, hash sets, snapshot conversion.
// Part of the device code of libsnacc_hip.so; see snk_common.hip.h for the execution model.
#pragma once
#include "snk_common.hip.h"

#include "snk_fast.hip.h"     // SNK_FSLOTS

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

// Exception granules (2-bit kernel on sequences with a few non-ACGT bytes, snk_fast.hip.h): one thread per
// 16-base granule; raw[g] is set when the granule holds a byte outside {A,C,G,T}; *count += flagged granules.
// lcase: 0x20 for a lower-case set (its letters are acgt), else 0.
__global__ void snk_excraw_kernel(const uint8_t *bytes, uint64_t n, uint32_t *raw, uint32_t *count, uint32_t lcase)
{
    const uint64_t ngran = (n + 15u) >> 4;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t mine = 0;
    for (; g < ngran; g += stride) {
        bool bad = false;
        for (uint32_t b = 0; b < 16u; ++b) {
            const uint64_t i = g * 16u + b;
            if (i < n) { const uint32_t c = bytes[i]; bad |= !(c == ('A' | lcase) || c == ('C' | lcase) || c == ('G' | lcase) || c == ('T' | lcase)); }
        }
        if (bad) { atomicOr(&raw[g >> 5], 1u << (g & 31u)); mine++; }
    }
    if (mine) atomicAdd(count, mine);
}

// Dilation by one granule either side: out bit g = raw bit g-1 | g | g+1, i.e. "an exception lies within
// bases [16g - 16, 16g + 32)" -- any 16-base window that starts in granule g is then covered by bit g alone.
__global__ void snk_excdilate_kernel(const uint32_t *raw, uint32_t nwords, uint32_t *out)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const uint32_t v = raw[w];
    const uint32_t lo = w ? raw[w - 1u] >> 31 : 0u, hi = w + 1u < nwords ? raw[w + 1u] << 31 : 0u;
    out[w] = v | (v << 1) | (v >> 1) | lo | hi;
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

// The mask arena of the 2-bit kernel (sequences with exceptions): same layout as the packed bytes, one 2-bit CLASS per base:
// 00 one of the set's four letters (ACGT, or acgt in a lower-case set) and behind the end, 01 the same letter in the OTHER
// case (soft-masked stretches), 11 any other byte (N, IUPAC codes, ...).  The 2-bit code (c >> 1) & 3 is the same for both
// cases of a letter, so (code, class) names the byte exactly unless the class is 11.
__global__ void snk_packmask_kernel(const uint8_t *bytes, uint64_t n, uint8_t *mask, uint32_t lcase)
{
    uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nbytes = (n + 3) >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; o < nbytes; o += stride) {
        uint32_t v = 0;
        for (uint32_t b = 0; b < 4u; ++b) {
            const uint64_t i = o * 4 + b;
            if (i < n) {
                const uint32_t c = bytes[i], u = c & ~0x20u;
                const bool letter = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
                if (!letter) v |= 3u << (2u * b);
                else if ((c & 0x20u) != lcase) v |= 1u << (2u * b);
            }
        }
        mask[o] = (uint8_t)v;
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

// The slot stream of the byte kernels (linked mode): slot[p] = table slot of the 5 bytes at p -- liblz4's 12-bit hash, or
// its renaming through the compact LUT -- for every position of a resident sequence (0 where fewer than 5 bytes are left:
// never probed).  Computed once per upload, it takes the 64-bit multiply and the hash -> slot LUT off the dependent
// chain of every probe: the tight loop loads the slots of cur-2 and cur with one 8-byte load.
__global__ void snk_slotstream_kernel(const uint8_t *bytes, uint64_t n, const uint16_t *lut /* NULL: slot = hash */, uint16_t *slots)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; p < n; p += stride) {
        uint32_t v = 0u;
        if (p + 5u <= n) {
            uint64_t w = 0;
            for (uint32_t b = 0; b < 5u; ++b) w |= (uint64_t)bytes[p + b] << (8u * b);
            const uint32_t h = (uint32_t)(((w << 24) * 889523592379ull) >> 52);
            v = lut ? lut[h] : h;
        }
        slots[p] = (uint16_t)v;
    }
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

// =========================================================================
//  segmented forms: ONE launch over many sequences (round 4)
// =========================================================================
// The per-sequence kernels above cost one launch per sequence and pass -- ~5 000 launches and copies for 1024 genomes,
// a fifth of the upload's wall time.  Here blockIdx.y names the sequence (entry y0 + blockIdx.y of the launch's id list,
// or the sequence of that number when ids == NULL) and the blocks of a row stride over it; every thread handles one
// 16-base granule with one 16-byte load (sequences start 64-byte aligned in the ASCII arena, which is zero behind their ends).
struct SnkSeqDesc {
    uint32_t boff;      // byte offset in the ASCII arena
    uint32_t len;
    uint32_t foff;      // first word of the sequence's granule flags
    uint32_t poff;      // byte offset in the packed / class arena (packed sequences only)
};

__device__ __forceinline__ uint32_t snk_seg_id(const uint32_t *ids, uint32_t y0)
{
    const uint32_t k = y0 + blockIdx.y;
    return ids ? ids[k] : k;
}

// 4 ASCII letters -> their four 2-bit codes ((c >> 1) & 3) in one byte, first letter lowest
__device__ __forceinline__ uint32_t snk_pack4(uint32_t w)
{
    const uint32_t t = (w >> 1) & 0x03030303u;
    return (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xFFu;
}

// 4 bytes -> their four 2-bit classes (00 a letter of the set's case, 01 of the other case, 11 another byte)
__device__ __forceinline__ uint32_t snk_class4(uint32_t w, uint32_t lcase)
{
    uint32_t v = 0u;
    for (uint32_t b = 0; b < 4u; ++b) {
        const uint32_t c = (w >> (8u * b)) & 255u, u = c & ~0x20u;
        const bool letter = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
        if (!letter) v |= 3u << (2u * b);
        else if ((c & 0x20u) != lcase) v |= 1u << (2u * b);
    }
    return v;
}

// snk_excraw_kernel for many sequences: raw[foff + g / 32] bit g % 32 = granule g holds a byte that is not one of the
// set's four letters; count[sequence] += flagged granules.
__global__ void snk_excraw_seg_kernel(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0,
                                      uint32_t *raw, uint32_t *count, uint32_t lcase)
{
    const uint32_t s = snk_seg_id(ids, y0);
    const SnkSeqDesc d = desc[s];
    const uint32_t ngran = (d.len + 15u) >> 4;
    const uint4 *src = (const uint4 *)(arena + d.boff);
    const uint32_t la = 'A' | lcase, lc = 'C' | lcase, lg = 'G' | lcase, lt = 'T' | lcase;
    uint32_t mine = 0u;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngran; g += gridDim.x * blockDim.x) {
        const uint4 v = src[g];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
        const uint32_t left = d.len - 16u * g, nv = left < 16u ? left : 16u;
        bool bad = false;
        for (uint32_t b = 0; b < 16u; ++b) {
            const uint32_t c = (w[b >> 2] >> (8u * (b & 3u))) & 255u;
            bad |= (b < nv) & !((c == la) | (c == lc) | (c == lg) | (c == lt));
        }
        if (bad) { atomicOr(&raw[d.foff + (g >> 5)], 1u << (g & 31u)); mine++; }
    }
    if (mine) atomicAdd(&count[s], mine);
}

// snk_excdilate_kernel for many sequences (fwords = words of the sequence's flags: ((len + 15) / 16 + 31) / 32 + 2)
__global__ void snk_excdilate_seg_kernel(const uint8_t * /* arena: unused */, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0,
                                         const uint32_t *raw, uint32_t *out)
{
    const uint32_t s = snk_seg_id(ids, y0);
    const SnkSeqDesc d = desc[s];
    const uint32_t nwords = (((d.len + 15u) >> 4) + 31u) / 32u + 2u;
    const uint32_t *r = raw + d.foff;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x) {
        const uint32_t v = r[w];
        const uint32_t lo = w ? r[w - 1u] >> 31 : 0u, hi = w + 1u < nwords ? r[w + 1u] << 31 : 0u;
        out[d.foff + w] = v | (v << 1) | (v >> 1) | lo | hi;
    }
}

// snk_pack_kernel / snk_packmask_kernel for many sequences: 16 bases -> one 32-bit word of the packed (MASK: class) arena
template <bool MASK>
__device__ __forceinline__ void snk_pack_seg_body(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0,
                                                  uint8_t *packed, uint32_t lcase)
{
    const uint32_t s = snk_seg_id(ids, y0);
    const SnkSeqDesc d = desc[s];
    const uint32_t ngran = (d.len + 15u) >> 4;
    const uint4 *src = (const uint4 *)(arena + d.boff);
    uint32_t *dst = (uint32_t *)(packed + d.poff);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngran; g += gridDim.x * blockDim.x) {
        uint4 v = src[g];
        if (MASK) {
            // (behind the end the arena holds zero bytes: class 00 there, as the per-sequence kernel writes it)
            const uint32_t left = d.len - 16u * g;
            uint32_t w[4] = { v.x, v.y, v.z, v.w };
            if (left < 16u)
                for (uint32_t b = left; b < 16u; ++b) w[b >> 2] = (w[b >> 2] & ~(255u << (8u * (b & 3u)))) | ((uint32_t)('A' | lcase) << (8u * (b & 3u)));
            dst[g] = snk_class4(w[0], lcase) | (snk_class4(w[1], lcase) << 8) | (snk_class4(w[2], lcase) << 16) | (snk_class4(w[3], lcase) << 24);
        } else {
            dst[g] = snk_pack4(v.x) | (snk_pack4(v.y) << 8) | (snk_pack4(v.z) << 16) | (snk_pack4(v.w) << 24);
        }
    }
}
__global__ void snk_pack_seg_kernel(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0, uint8_t *packed)
{
    snk_pack_seg_body<false>(arena, desc, ids, y0, packed, 0u);
}
__global__ void snk_packmask_seg_kernel(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0, uint8_t *mask, uint32_t lcase)
{
    snk_pack_seg_body<true>(arena, desc, ids, y0, mask, lcase);
}

// snk_hashset_kernel / snk_hashset4_kernel for many sequences (H4: the one-shot hash of 4 bytes, 256 words; else 128)
template <bool H4>
__device__ __forceinline__ void snk_hashset_seg_body(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0, uint32_t *set)
{
    constexpr uint32_t W = H4 ? 256u : 128u, K = H4 ? 4u : 5u;
    __shared__ uint32_t local[W];
    for (uint32_t t = threadIdx.x; t < W; t += blockDim.x) local[t] = 0u;
    __syncthreads();
    const uint32_t s = snk_seg_id(ids, y0);
    const SnkSeqDesc d = desc[s];
    const uint8_t *bytes = arena + d.boff;
    if (d.len >= K) {
        const uint32_t last = d.len - K;
        for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p <= last; p += gridDim.x * blockDim.x) {
            uint64_t v = 0;
            for (uint32_t b = 0; b < K; ++b) v |= (uint64_t)bytes[p + b] << (8u * b);
            const uint32_t h = H4 ? snk_hash4(v) : snk_hash5(v);
            atomicOr(&local[h >> 5], 1u << (h & 31u));
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < W; t += blockDim.x)
        if (local[t]) atomicOr(&set[t], local[t]);
}
__global__ void snk_hashset_seg_kernel(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0, uint32_t *set)
{
    snk_hashset_seg_body<false>(arena, desc, ids, y0, set);
}
__global__ void snk_hashset4_seg_kernel(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0, uint32_t *set)
{
    snk_hashset_seg_body<true>(arena, desc, ids, y0, set);
}

// snk_slotstream_kernel for many sequences (slots: the stream of the whole ASCII arena, same offsets)
__global__ void snk_slotstream_seg_kernel(const uint8_t *arena, const SnkSeqDesc *desc, const uint32_t *ids, uint32_t y0,
                                          const uint16_t *lut /* NULL: slot = hash */, uint16_t *slots)
{
    const uint32_t s = snk_seg_id(ids, y0);
    const SnkSeqDesc d = desc[s];
    const uint8_t *bytes = arena + d.boff;
    uint16_t *out = slots + d.boff;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < d.len; p += gridDim.x * blockDim.x) {
        uint32_t v = 0u;
        if (p + 5u <= d.len) {
            uint64_t w = 0;
            for (uint32_t b = 0; b < 5u; ++b) w |= (uint64_t)bytes[p + b] << (8u * b);
            const uint32_t h = snk_hash5(w);
            v = lut ? lut[h] : h;
        }
        out[p] = (uint16_t)v;
    }
}
