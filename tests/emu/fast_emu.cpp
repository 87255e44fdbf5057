// fast_emu.cpp -- TEST INFRASTRUCTURE.  Runs the 2-bit kernel's source (snk_fast.hip.h) on the CPU,
// one lane at a time (snk_host_emu.h), over the same HBM layout snk_upload builds: packed arena,
// ASCII arena, 5-mer -> slot LUT, exception flags, prefix snapshots from the singles pass, then the
// ordered pairs.  Sequences with a few non-ACGT bytes run the instantiation for exceptions (EXC).
// Build: g++ -O1 -DSNK_HOST_EMU -I tests/emu -I snacc_amd/csrc -shared -fPIC -o tests/emu/libfast_emu.so tests/emu/fast_emu.cpp
#include "snk_fast.hip.h"
#include <vector>

namespace {

uint32_t host_hash5(const uint8_t *p)
{
    uint64_t v = 0;
    memcpy(&v, p, 5);
    return (uint32_t)(((v << 24) * 889523592379ull) >> 52);
}

uint8_t g_lcase = 0;      // 0x20: the set's letters are acgt (as snk_upload decides by the majority)
bool acgt(uint8_t c) { return c == ('A' | g_lcase) || c == ('C' | g_lcase) || c == ('G' | g_lcase) || c == ('T' | g_lcase); }

} // namespace

// exc_limit: flagged 16-base granules per 2^20 bases (+8) up to which a sequence with non-ACGT bytes stays on
// the 2-bit kernel (0: pure ACGT only).  Entries of `singles` / `pairs` that the kernel does not serve stay 0.
// far != 0: the pairs run as a FAR chain (table in "global" memory: u32 absolute positions, snk_fast_wave<.., FAR = true>)
// -- only for sets without exceptions (as the launch code decides).
// spec != 0: two lanes per chain (snk_fast_wave<.., SPEC = true>, what every product launch of the 2-bit kernel runs): the
// emulated lane is role 0 of its pair, the partner runs on a second host thread inside the steady loop (snk_host_emu.h).
template <bool EXC> static void emu_launch(const SnkTables &T, const SnkFastGrid &G, uint32_t *out, uint32_t *status, bool spec)
{
    if (!spec) { snk_fast_kernel_body<false, EXC>(T, G, 1u, out, status); return; }
    for (uint32_t t = 0; t < 512u; ++t) ((uint32_t *)snk_lds8)[t] = ((const uint32_t *)T.lut_slot)[t];
    if (EXC && G.flut > SNK_FLUT_B) for (uint32_t t = 0; t < 512u; ++t) ((uint32_t *)snk_lds8)[512u + t] = ((const uint32_t *)T.lut_oj)[t];
    snk_fast_wave<false, EXC, false, true>(T, G, 1u, out, status);
}

extern "C" int emu_fast_sizes(int n, const uint8_t *const *seqs, const uint64_t *lens,
                              uint32_t *singles, uint32_t *pairs, uint32_t header_bytes, uint32_t exc_limit, uint32_t lower, uint32_t far,
                              uint32_t spec)
{
    g_lcase = lower ? 0x20 : 0;
    const char code2byte[4] = { (char)('A' | g_lcase), (char)('C' | g_lcase), (char)('T' | g_lcase), (char)('G' | g_lcase) };
    std::vector<uint16_t> slot(1024, 0), h2s(4096, 0xFFFF), s2h(SNK_FSLOTS, 0);
    {
        std::vector<int> slot_of_hash(4096, -1);
        int n_slots = 0;
        for (uint32_t k = 0; k < 1024; ++k) {
            uint8_t b[5];
            for (int i = 0; i < 5; ++i) b[i] = (uint8_t)code2byte[(k >> (2 * i)) & 3];
            const uint32_t h = host_hash5(b);
            if (slot_of_hash[h] < 0) slot_of_hash[h] = n_slots++;
            slot[k] = (uint16_t)slot_of_hash[h];
            h2s[h] = slot[k]; s2h[slot[k]] = (uint16_t)h;
        }
        if (n_slots >= (int)SNK_FSLOTS) return -1;         // (the last slot stays free: "nothing owed" puts go there)
    }
    // the other case's 5-mers: their own compact numbering (oj: code -> j; the other-case mode keeps its table in LDS:
    // snk_oth_swap_in), which is also where their hashes sit in a chain's overflow table (ovi: hash -> index; the other hashes
    // follow); okey: shared slot of the 2-bit table, or 0x1000 | overflow index, for the general path; both LUTs back to back as
    // the kernels for sequences with exceptions hold them in LDS
    std::vector<uint16_t> okey(1024, 0), oj(1024, 0), ovi(4096, 0xFFFF);
    std::vector<uint32_t> omap(SNK_FSLOTS, 0xFFFF0000u);
    {
        std::vector<uint32_t> oh(1024);
        uint32_t ncls = 0;
        for (uint32_t k = 0; k < 1024; ++k) {
            uint8_t b[5];
            for (int i = 0; i < 5; ++i) b[i] = (uint8_t)(code2byte[(k >> (2 * i)) & 3] ^ 0x20);
            const uint32_t h = host_hash5(b);
            oh[k] = h;
            if (ovi[h] == 0xFFFF) ovi[h] = (uint16_t)ncls++;
            oj[k] = ovi[h];
        }
        if (ncls >= SNK_FSLOTS) return -1;
        for (uint32_t h = 0; h < 4096; ++h) if (ovi[h] == 0xFFFF) ovi[h] = (uint16_t)ncls++;
        for (uint32_t k = 0; k < 1024; ++k) {
            const bool shared = h2s[oh[k]] != 0xFFFF;
            okey[k] = shared ? h2s[oh[k]] : (uint16_t)(0x1000u | oj[k]);
            omap[oj[k]] = oh[k] | ((uint32_t)h2s[oh[k]] << 16);
        }
        for (uint32_t h = 0; h < 4096; ++h) if (h2s[h] == 0xFFFF) h2s[h] = (uint16_t)(0x8000u | ovi[h]);
    }
    std::vector<uint8_t> ok((size_t)n, 0), exc((size_t)n, 0);
    std::vector<uint32_t> poff((size_t)n, 0), boff((size_t)n, 0), len((size_t)n), spos((size_t)n), eoff((size_t)n, 0xFFFFFFFFu);
    std::vector<std::vector<uint32_t>> flags((size_t)n);
    size_t ptot = SNK_ARENA_SLACK, btot = SNK_PAD, ftot = 0;
    bool any_exc = false;
    for (int g = 0; g < n; ++g) {
        len[g] = (uint32_t)lens[g];
        boff[g] = (uint32_t)btot; btot += ((size_t)lens[g] + 63) / 64 * 64 + SNK_PAD;
        // raw flags per 16-base granule, then the dilation (as snk_excraw_kernel / snk_excdilate_kernel)
        const size_t ngran = ((size_t)lens[g] + 15) / 16, nwords = (ngran + 31) / 32 + 2;
        std::vector<uint32_t> raw(nwords, 0);
        uint32_t cnt = 0;
        for (size_t q = 0; q < ngran; ++q) {
            bool bad = false;
            for (size_t i = q * 16; i < q * 16 + 16 && i < lens[g]; ++i) bad |= !acgt(seqs[g][i]);
            if (bad) { raw[q >> 5] |= 1u << (q & 31); cnt++; }
        }
        flags[g].assign(nwords, 0);
        for (size_t w = 0; w < nwords; ++w) {
            const uint32_t v = raw[w], lo = w ? raw[w - 1] >> 31 : 0u, hi = w + 1 < nwords ? raw[w + 1] << 31 : 0u;
            flags[g][w] = v | (v << 1) | (v >> 1) | lo | hi;
        }
        const uint64_t allowed = 8u + lens[g] * (uint64_t)exc_limit / 1048576u;
        ok[g] = lens[g] > 0 && (cnt == 0 || (exc_limit > 0 && cnt <= allowed));
        exc[g] = ok[g] && cnt != 0;
        if (exc[g]) { any_exc = true; eoff[g] = (uint32_t)ftot; }
        ftot += nwords;
        if (ok[g]) { poff[g] = (uint32_t)ptot; ptot += (((size_t)lens[g] + 3) / 4 + 63) / 64 * 64 + SNK_PAD; }
        spos[g] = lens[g] > SNK_BLOCK ? (uint32_t)(lens[g] / SNK_BLOCK * SNK_BLOCK) : 0u;
    }
    ptot += SNK_ARENA_SLACK;
    std::vector<uint8_t> arena(ptot, 0), marena(ptot, 0), bytes(btot + SNK_PAD, 0), zero(4 * SNK_PAD, 0);
    std::vector<uint32_t> fl(ftot + 1, 0);
    {
        size_t f = 0;
        for (int g = 0; g < n; ++g) {
            memcpy(bytes.data() + boff[g], seqs[g], lens[g]);
            if (exc[g]) memcpy(fl.data() + f, flags[g].data(), flags[g].size() * 4);
            f += flags[g].size();
            if (!ok[g]) continue;
            for (uint64_t i = 0; i < lens[g]; ++i)
                arena[poff[g] + (i >> 2)] |= (uint8_t)(((seqs[g][i] >> 1) & 3u) << (2u * (i & 3u)));
            for (uint64_t i = 0; i < lens[g]; ++i)          // class arena: 00 the set's letters, 01 the other case, 11 another byte
                if (!acgt(seqs[g][i])) {
                    const uint8_t u = (uint8_t)(seqs[g][i] & ~0x20u);
                    const bool letter = u == 'A' || u == 'C' || u == 'G' || u == 'T';
                    marena[poff[g] + (i >> 2)] |= (uint8_t)((letter ? 1u : 3u) << (2u * (i & 3u)));
                }
        }
    }
    std::vector<uint32_t> snap_out((size_t)n, 0), snap_fast((size_t)n * SNK_FSLOTS, 0), snap_gen((size_t)n * 4096, 0), status(1, 0);
    std::vector<uint32_t> ovf(4096, 0), osave(512, 0);
    // exact runs of non-ACGT bytes of the sequences with exceptions (as snk_upload builds them)
    std::vector<uint32_t> runs, roff((size_t)n, 0);
    for (int g = 0; g < n; ++g) {
        if (!exc[g]) continue;
        roff[g] = (uint32_t)(runs.size() / 2);
        for (uint64_t i = 0; i < lens[g]; ++i) {
            if (acgt(seqs[g][i])) continue;
            if (!runs.empty() && runs.size() / 2 > roff[g] && runs.back() == (uint32_t)i) runs.back() = (uint32_t)i + 1;
            else { runs.push_back((uint32_t)i); runs.push_back((uint32_t)i + 1); }
        }
        runs.push_back(0xFFFFFFFFu); runs.push_back(0xFFFFFFFFu);
    }
    runs.push_back(0xFFFFFFFFu); runs.push_back(0xFFFFFFFFu);

    SnkTables T;
    memset(&T, 0, sizeof T);
    T.packed_arena = arena.data(); T.mask_arena = marena.data(); T.packed_off = poff.data(); T.len = len.data();
    T.bytes_arena = bytes.data(); T.bytes_off = boff.data(); T.zero_pad = zero.data();
    T.snap_pos = spos.data(); T.snap_out = snap_out.data(); T.snap_fast = snap_fast.data(); T.snap_gen = snap_gen.data();
    T.lut_oj = oj.data(); T.lut_omap = omap.data(); T.lut_ovi = ovi.data(); T.osave = osave.data();
    T.lut_slot = slot.data(); T.lut_h2s = h2s.data(); T.lut_s2h = s2h.data(); T.lut_okey = okey.data(); T.header_bytes = header_bytes;
    T.exc_flags = fl.data(); T.exc_off = eoff.data(); T.ovf = ovf.data(); T.exc_runs = runs.data(); T.exc_roff = roff.data();

    blockDim.x = 64; threadIdx.x = 0; blockIdx.x = 0;
    for (int g = 0; g < n; ++g) {
        singles[g] = 0;
        if (!ok[g] || lens[g] <= SNK_BLOCK) continue;
        SnkJob jb; jb.xi = g; jb.yi = -1; jb.out_idx = (uint32_t)g; jb.snap = spos[g] ? 1 : 0;
        SnkFastGrid G; G.jobs = &jb; G.n_jobs = 1u; G.r0 = 0u; G.rows = 1u; G.n = 1u; G.batch = 1u; G.queue = nullptr; G.yorder = nullptr;
        if (any_exc) G.flut = 2u * SNK_FLUT_B;
        if (any_exc) emu_launch<true>(T, G, singles, status.data(), spec != 0);
        else         emu_launch<false>(T, G, singles, status.data(), spec != 0);
    }
    // all eligible pairs as ONE launch of one lane: the lane walks the job list through the kernel's own
    // hand-out loop (a finished lane takes the next job), alternately as an explicit list and, when every
    // pair of the set is eligible, as a dense tile
    std::vector<SnkJob> list;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            pairs[(size_t)i * n + j] = 0;
            if (!ok[i] || !ok[j] || lens[i] + lens[j] <= SNK_BLOCK) continue;
            SnkJob jb; jb.xi = i; jb.yi = j; jb.out_idx = (uint32_t)((size_t)i * n + j); jb.snap = 0;
            list.push_back(jb);
        }
    if (!list.empty()) {
        SnkFastGrid G; G.r0 = 0u; G.rows = (uint32_t)n; G.n = (uint32_t)n; G.batch = 3u;
        if (any_exc) G.flut = 2u * SNK_FLUT_B;
        std::vector<uint32_t> order((size_t)n);
        for (int k = 0; k < n; ++k) order[(size_t)k] = (uint32_t)(n - 1 - k);
        G.yorder = (n & 2) ? order.data() : nullptr;
        uint32_t counter = 0u;                           // the queue counts the jobs handed out
        G.queue = (n & 1) ? &counter : nullptr;          // both schedules get exercised
        if (list.size() == (size_t)n * n) { G.jobs = nullptr; G.n_jobs = (uint32_t)(n * n); }
        else                              { G.jobs = list.data(); G.n_jobs = (uint32_t)list.size(); }
        std::vector<uint32_t> far_tab(SNK_FSLOTS, 0xDEADBEEFu);
        if (far && !any_exc) { G.far_tab = far_tab.data(); G.lds_waves = 0u; G.far_lanes = 1u; G.far_stop = 0u; G.queue = &counter; }
        if (any_exc)                   emu_launch<true>(T, G, pairs, status.data(), spec != 0);
        else if (far)                  snk_fast_kernel_body<false, false>(T, G, 1u, pairs, status.data());
        else                           emu_launch<false>(T, G, pairs, status.data(), spec != 0);
    }
    return (int)status[0];
}


extern "C" unsigned long long emu_other_mode_trips(void) { return snk_emu_oth_trips; }
