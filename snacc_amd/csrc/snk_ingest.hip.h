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
