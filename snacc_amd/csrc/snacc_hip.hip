// snacc_hip.hip -- C-ABI (include/snacc_hip.h) over the gfx950 kernels in
// snk_device.hip.h.  Host logic only: device memory, job lists, launches.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -I../../include \
//               -o ../libsnacc_hip.so snacc_hip.hip
#include "snk_device.hip.h"
#include "snacc_hip.h"
#include "snk_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct snk_ctx_impl {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;      // event pair of the last launch (owned by ev_log)
    bool ev_valid = false;
    double ms_accum = -1.0;          // sum over the tiles of the last snk_pairs call
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_log;   // one pair per pair launch since the last snk_pairs_ms_log
    size_t ev_used = 0;
    bool ev_logging = false;         // a caller reads the log (snk_pairs_ms_log): keep one event pair per launch; else ONE pair, reused
    int n_cus = 0;                   // compute units of the device: size of a persistent launch
    uint32_t *d_yorder = nullptr;    // sequence indices by decreasing length (column order of ragged dense tiles)
    uint32_t *d_queue = nullptr;     // ring of batch counters for launches with the dynamic schedule
    unsigned queue_next = 0;
    int fast_dynamic = -1;           // -1 auto (by length spread), 0 static round robin, 1 atomic queue
    bool dense_tile = false;         // the last build_jobs found every pair of the tile fit for the 2-bit kernel
    hipEvent_t jobs_busy = nullptr;  // last launch that reads d_jobs: waited for before the list is rewritten
    uint32_t *d_far = nullptr; size_t far_bytes = 0;     // tables of the 2-bit kernel's far chains (one set per context)
    hipEvent_t far_busy = nullptr; bool far_in_flight = false;   // ... in use until the launch that got them ends
    uint32_t *d_bgt = nullptr; size_t bgt_bytes = 0;     // global-memory tables of the byte kernels ("bytes_gt"), same rules
    hipEvent_t bgt_busy = nullptr; bool bgt_in_flight = false;
    int bytes_spec = 0;                                  // LDS byte kernels on the slot stream: two lanes per chain (snk_bytes_loop2_spec)
    int bytes_gt = -1, bytes_gt_wgs = 1;                 // waves per workgroup (0 = tables in LDS, -1 = choose), workgroups per CU and launch
    int far_lanes = 0, far_waves = 4; // far_lanes 0 = no far chains
    int far_min = 4;                 // far chains only in launches of at least far_min jobs per LDS chain of the card (0: always; tests)
    int far_stop_pct = 140;          // far waves take no new jobs once fewer than this % of (LDS chains of the launch) jobs are left
    hipEvent_t ovf_busy = nullptr;   // last 2-bit launch with exceptions: its chains' overflow tables (one set per context) are
    bool ovf_in_flight = false;      // in use until it ends -- the next such launch waits for it on its own stream
    std::string err;

    // options
    bool fast_asm = true;            // 0 = the C++ statement of the 2-bit kernel's steady loop (cross-checks)
    int fast_spec = 1;               // pair launches run with two lanes per chain (snk_fast_steady_spec); 0 = one lane; 3 = three
                                     // lanes, C++ statement (snk_fast_steady_spec3: round 4 experiment, pure-ACGT sets)
    int fast_lanes = 0, fast_waves = 4, gen_chains = 8, bytes_lanes = 9, bytes_waves = 2;   // fast_lanes 0 = as many as the LDS holds
    int cbytes_lanes = 17, cbytes_waves = 4;   // compact byte kernel, 1024 slots: up to 70 chains per CU
    int c2bytes_lanes = 17, c2bytes_waves = 2; // compact byte kernel, 2048 slots: up to 35 chains per CU
    int compact_cap = 0;                       // 0 / 1024 / 2048
    int os_lanes = 9, os_waves = 1;            // one-shot kernel, full 8192-slot table: 9 chains per CU
    int cos_lanes = 16, cos_waves = 4;         // one-shot kernel, compact table: up to 67 chains per CU
    bool oneshot_compact_ok = false; int n_hashes4 = 0;
    int bytes_compact_opt = -1;                // -1 auto, 0 never, 1 whenever the resident hash set allows
    bool compact_ok = false;                   // the resident sequences use <= 2048 distinct 5-byte hashes
    int n_hashes = 0;
    bool bytes_legacy = false;      // 1 = linked-mode byte jobs also go to the legacy u32-table kernel
    bool force_generic = false;
    uint32_t header_bytes = 7;
    uint64_t arena_limit = 0xFFF00000ull;   // arenas are addressed with 32-bit offsets; option arena_limit lowers it (tests)

    // resident sequences
    int n = 0, n_packed = 0;
    uint32_t min_len = 0, max_len = 0;
    std::vector<uint32_t> len;
    std::vector<uint32_t> boff;      // byte offset of every sequence in d_bytes
    std::vector<uint8_t> is_packed;  // goes to the 2-bit kernel: pure ACGT, or ACGT with a few exceptions
    std::vector<uint8_t> has_exc;    // ... the latter
    bool any_exc = false;
    bool any_other = false;          // ... and some exception byte of a 2-bit sequence is a letter of the other case (soft-masked stretches)
    bool lower = false;              // the resident set's letters are acgt: its 2-bit sequences, LUTs and exceptions go by the lower case
    long exc_limit = 2048;           // a sequence stays on the 2-bit kernel up to 4 + 1.25 * exc_limit exception sites (runs of bytes that
                                     // are not the set's letters) per 2^20 bases (see snk_upload); beyond that the byte kernels are
                                     // faster (measured crossover: 2400 IUPAC sites per Mbp; soft-masking: never, round 4)
    uint32_t *d_exc_flags = nullptr, *d_exc_off = nullptr, *d_ovf = nullptr; size_t ovf_bytes = 0;
    uint32_t *d_exc_runs = nullptr, *d_exc_roff = nullptr;
    uint16_t *d_lut_h2s = nullptr, *d_lut_s2h = nullptr, *d_lut_okey = nullptr;
    uint16_t *d_lut_oj = nullptr, *d_lut_ovi = nullptr; uint32_t *d_lut_omap = nullptr;     // the other-case mode's compact slots (snk_oth_swap_in)
    uint32_t *d_osave = nullptr;                                      // ... and its save areas, one per resident chain (beside d_ovf)
    // deflate add-on (snk_deflate.hip): opaque state + its destructor
    void *dfl = nullptr; void (*dfl_free)(void *) = nullptr;
    bool dfl_serial = false, dfl_kmer = true, dfl_norestart = false;
    uint8_t *d_bytes = nullptr, *d_packed = nullptr, *d_pmask = nullptr, *d_zero = nullptr;
    uint16_t *d_slots = nullptr;     // slot stream of the byte kernels (2 bytes per byte of the ASCII arena)
    const uint8_t **d_bytes_ptr = nullptr; uint32_t *d_packed_off = nullptr, *d_bytes_off = nullptr;
    uint32_t *d_len = nullptr, *d_snap_pos = nullptr, *d_snap_out = nullptr;
    uint32_t *d_snap_fast = nullptr, *d_snap_gen = nullptr;
    uint16_t *d_lut_slot = nullptr, *d_lut_h2c = nullptr, *d_lut_h2c4 = nullptr;
    uint32_t *d_lut_hash = nullptr, *d_hashset = nullptr;      // d_hashset: 128 words (hash5) + 256 words (hash4)
    uint32_t *d_single = nullptr, *d_status = nullptr;
    bool singles_done = false;       // snk_upload has completed: sequences resident (phase A may still be owed, see single_have)
    size_t n_fast_clean = 0;         // (build_jobs) how many of the list's 2-bit jobs pair two sequences WITHOUT exceptions: they come first
    int split_clean = 1;             // option: such pairs of a set with exceptions run on the pure kernel (0 never, 1 when they fill the card 16 times, 2 always)
    bool defer_singles = false;      // option: snk_upload leaves phase A (single sizes + prefix snapshots) to the calls that need it,
                                     // row by row -- a rank of a sharded run computes its own rows only, a gzip / zlib run none
    std::vector<uint8_t> single_have;   // per sequence: phase A done
    double up_ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // host wall time of the last upload's stages (snk_upload_times)

    // scratch for pair launches (grown on demand)
    SnkJob *d_jobs = nullptr; size_t jobs_cap = 0; bool jobs_in_flight = false;
    uint32_t *d_out = nullptr; size_t out_cap = 0;
    std::vector<SnkJob> h_jobs;
};

int fail(snk_ctx_impl *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), SNK_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

template <typename T> void dfree(T *&p) { if (p) { (void)hipFree((void *)p); p = nullptr; } }

// device temporaries of one call: freed on every return path
template <typename T> struct DevTemp {
    T *p = nullptr;
    ~DevTemp() { if (p) (void)hipFree((void *)p); }
    void release() { if (p) { (void)hipFree((void *)p); p = nullptr; } }
};

// ---- host-to-device copy of the ASCII arena through pinned staging buffers ------------------------------------------
// The caller's sequences are pageable memory: one hipMemcpy per sequence moves 1 GB in ~55-85 ms (1024 calls, the runtime's
// own staging at 12-18 GB/s).  Here SNK_STAGE_THREADS host threads build the arena image -- the sequences back to back at
// their offsets, the zero padding between them included -- chunk by chunk in pinned buffers (two per thread, so that a
// thread fills one while the DMA engine drains the other) and send every chunk with one asynchronous copy on the thread's own
// stream.  Buffers, streams and events are a process-wide pool (allocated at the first upload, per device).
#define SNK_STAGE_THREADS 6
#define SNK_STAGE_CHUNK   ((size_t)4 << 20)
struct SnkStagePool {
    std::mutex mu;                 // one staged upload at a time per process (contexts are independent otherwise)
    int device = -1;
    void *buf[SNK_STAGE_THREADS][2] = {};
    hipStream_t st[SNK_STAGE_THREADS] = {};
    hipEvent_t ev[SNK_STAGE_THREADS][2] = {};
};
SnkStagePool g_stage;

// (called with g_stage.mu held)
hipError_t stage_pool_prepare(int device)
{
    SnkStagePool &P = g_stage;
    if (P.device == device) return hipSuccess;
    for (int t = 0; t < SNK_STAGE_THREADS; ++t) {          // another device (or first use): rebuild
        for (int b = 0; b < 2; ++b) {
            if (P.buf[t][b]) { (void)hipHostFree(P.buf[t][b]); P.buf[t][b] = nullptr; }
            if (P.ev[t][b]) { (void)hipEventDestroy(P.ev[t][b]); P.ev[t][b] = nullptr; }
        }
        if (P.st[t]) { (void)hipStreamDestroy(P.st[t]); P.st[t] = nullptr; }
    }
    P.device = -1;
    for (int t = 0; t < SNK_STAGE_THREADS; ++t) {
        hipError_t e = hipStreamCreateWithFlags(&P.st[t], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        for (int b = 0; b < 2; ++b) {
            e = hipHostMalloc(&P.buf[t][b], SNK_STAGE_CHUNK, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&P.ev[t][b], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
    }
    P.device = device;
    return hipSuccess;
}

// d_bytes[0, btot) = the arena image of the sequences (seqs[g] at boff[g], zeros elsewhere).  Blocking.
hipError_t stage_upload(int device, uint8_t *d_bytes, size_t btot, size_t n, const uint8_t *const *seqs, const uint64_t *lens,
                        const std::vector<size_t> &boff)
{
    std::lock_guard<std::mutex> lock(g_stage.mu);
    hipError_t e0 = stage_pool_prepare(device);
    if (e0 != hipSuccess) return e0;
    const size_t nchunks = (btot + SNK_STAGE_CHUNK - 1) / SNK_STAGE_CHUNK;
    const int nt = (int)std::min<size_t>(SNK_STAGE_THREADS, nchunks);
    std::atomic<int> err((int)hipSuccess);
    auto work = [&](int t) {
        SnkStagePool &P = g_stage;
        if (hipSetDevice(device) != hipSuccess) { err = (int)hipErrorInvalidDevice; return; }
        bool used[2] = { false, false };
        int b = 0;
        for (size_t k = (size_t)t; k < nchunks && err.load() == (int)hipSuccess; k += (size_t)nt, b ^= 1) {
            const size_t a = k * SNK_STAGE_CHUNK, e = std::min(btot, a + SNK_STAGE_CHUNK);
            if (used[b]) { const hipError_t r = hipEventSynchronize(P.ev[t][b]); if (r != hipSuccess) { err = (int)r; return; } }
            uint8_t *dst = (uint8_t *)P.buf[t][b];
            // first sequence whose slot reaches into [a, e): slots are in offset order
            size_t g = (size_t)(std::upper_bound(boff.begin(), boff.end(), a) - boff.begin());
            if (g > 0) --g;
            size_t pos = a;                                    // arena bytes [a, pos) of the chunk are written
            for (; g < n && boff[g] < e; ++g) {
                const size_t s0 = boff[g], s1 = boff[g] + (size_t)lens[g];
                const size_t c0 = std::max(s0, pos), c1 = std::min(s1, e);
                if (c1 <= c0) continue;
                if (c0 > pos) memset(dst + (pos - a), 0, c0 - pos);
                memcpy(dst + (c0 - a), seqs[g] + (c0 - s0), c1 - c0);
                pos = c1;
            }
            if (e > pos) memset(dst + (pos - a), 0, e - pos);
            hipError_t r = hipMemcpyAsync(d_bytes + a, dst, e - a, hipMemcpyHostToDevice, P.st[t]);
            if (r == hipSuccess) r = hipEventRecord(P.ev[t][b], P.st[t]);
            if (r != hipSuccess) { err = (int)r; return; }
            used[b] = true;
        }
        const hipError_t r = hipStreamSynchronize(P.st[t]);
        if (r != hipSuccess) err = (int)r;
    };
    std::vector<std::thread> ts;
    for (int t = 1; t < nt; ++t) ts.emplace_back(work, t);
    work(0);
    for (auto &th : ts) th.join();
    return (hipError_t)err.load();
}

void free_sequences(snk_ctx_impl *c)
{
    dfree(c->d_bytes); dfree(c->d_packed); dfree(c->d_pmask); dfree(c->d_slots); dfree(c->d_bytes_ptr); dfree(c->d_packed_off); dfree(c->d_bytes_off);
    dfree(c->d_len); dfree(c->d_snap_pos); dfree(c->d_snap_out); dfree(c->d_snap_fast); dfree(c->d_yorder);
    dfree(c->d_snap_gen); dfree(c->d_single); dfree(c->d_exc_flags); dfree(c->d_exc_off); dfree(c->d_ovf); dfree(c->d_osave); c->ovf_bytes = 0; c->ovf_in_flight = false;
    dfree(c->d_exc_runs); dfree(c->d_exc_roff);
    dfree(c->d_far); c->far_bytes = 0; c->far_in_flight = false;
    dfree(c->d_bgt); c->bgt_bytes = 0; c->bgt_in_flight = false;
    c->has_exc.clear(); c->any_exc = false; c->any_other = false;
    c->n = 0; c->n_packed = 0; c->len.clear(); c->boff.clear(); c->is_packed.clear(); c->singles_done = false; c->single_have.clear();
    if (c->dfl && c->dfl_free) c->dfl_free(c->dfl);
    c->dfl = nullptr;
}

// ---- the 5-mer LUTs of the 2-bit kernel ---------------------------------------------------
// code = (byte >> 1) & 3  =>  A=0 C=1 T=2 G=3
const char kCodeToByte[4] = { 'A', 'C', 'T', 'G' };

uint32_t host_hash5(const uint8_t *p)
{
    uint64_t v = 0;
    memcpy(&v, p, 5);
    return (uint32_t)(((v << 24) * 889523592379ull) >> 52);
}

// lower: the LUTs of a lower-case set (the 2-bit code (c >> 1) & 3 is the same for a letter's two cases; liblz4's hash
// of the bytes is not: 894 slots for ACGT, 895 for acgt -- the last slot of the table stays free for "nothing owed")
bool build_luts(std::vector<uint16_t> &slot, std::vector<uint32_t> &hash, bool lower = false)
{
    slot.assign(1024, 0); hash.assign(1024, 0);
    std::vector<int> slot_of_hash(4096, -1);
    int n_slots = 0;
    for (uint32_t k = 0; k < 1024; ++k) {
        uint8_t b[5];
        for (int i = 0; i < 5; ++i) b[i] = (uint8_t)(kCodeToByte[(k >> (2 * i)) & 3] | (lower ? 0x20 : 0));
        hash[k] = host_hash5(b);
        if (slot_of_hash[hash[k]] < 0) slot_of_hash[hash[k]] = n_slots++;
        slot[k] = (uint16_t)slot_of_hash[hash[k]];
    }
    return n_slots < (int)SNK_FSLOTS;
}

// The 2-bit kernel's LUTs for the context's letter case, to the device.
int upload_luts(snk_ctx_impl *c)
{
    std::vector<uint16_t> slot; std::vector<uint32_t> hash;
    if (!build_luts(slot, hash, c->lower)) return fail(c, SNK_E_STATE, "5-mer slot count exceeds the table");
    std::vector<uint16_t> h2s(4096, 0xFFFF), s2h(SNK_FSLOTS, 0);       // hash <-> slot (general path of sequences with exceptions)
    for (uint32_t k = 0; k < 1024; ++k) { h2s[hash[k]] = slot[k]; s2h[slot[k]] = (uint16_t)hash[k]; }
    {   // the other case's 5-mers: their own compact numbering (oslot: code -> j), which is also where their hashes sit in a chain's
        // overflow table (ovi: hash -> index; the other hashes follow); shared slot, or the overflow index, for the general path
        std::vector<uint16_t> oslot; std::vector<uint32_t> ohash;
        build_luts(oslot, ohash, !c->lower);
        std::vector<uint16_t> ovi(4096, 0xFFFF);
        uint32_t n_o = 0;
        for (uint32_t k = 0; k < 1024; ++k) { ovi[ohash[k]] = oslot[k]; n_o = std::max<uint32_t>(n_o, oslot[k] + 1u); }
        for (uint32_t h = 0; h < 4096; ++h) if (ovi[h] == 0xFFFF) ovi[h] = (uint16_t)n_o++;
        std::vector<uint16_t> okey(1024);
        std::vector<uint32_t> omap(SNK_FSLOTS, 0xFFFF0000u);
        for (uint32_t k = 0; k < 1024; ++k) {
            const bool shared = h2s[ohash[k]] != 0xFFFF;
            okey[k] = shared ? h2s[ohash[k]] : (uint16_t)(0x1000u | oslot[k]);
            omap[oslot[k]] = ohash[k] | ((uint32_t)h2s[ohash[k]] << 16);
        }
        for (uint32_t h = 0; h < 4096; ++h) if (h2s[h] == 0xFFFF) h2s[h] = (uint16_t)(0x8000u | ovi[h]);      // (after its use as "shared?" above)
        HIPCHK(c, hipMemcpy(c->d_lut_okey, okey.data(), 2048, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_lut_oj, oslot.data(), 2048, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_lut_omap, omap.data(), SNK_FSLOTS * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_lut_ovi, ovi.data(), 8192, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemcpy(c->d_lut_h2s, h2s.data(), 8192, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_lut_s2h, s2h.data(), SNK_FSLOTS * 2, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_lut_slot, slot.data(), 2048, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_lut_hash, hash.data(), 4096, hipMemcpyHostToDevice));
    return SNK_OK;
}

SnkTables make_tables(const snk_ctx_impl *c)
{
    SnkTables T;
    T.bytes = c->d_bytes_ptr; T.bytes_arena = c->d_bytes; T.bytes_off = c->d_bytes_off; T.packed_arena = c->d_packed; T.mask_arena = c->d_pmask; T.packed_off = c->d_packed_off; T.len = c->d_len;
    T.snap_pos = c->d_snap_pos; T.snap_out = c->d_snap_out;
    T.snap_fast = c->d_snap_fast; T.snap_gen = c->d_snap_gen;
    T.exc_runs = c->d_exc_runs; T.exc_roff = c->d_exc_roff;
    T.exc_flags = c->d_exc_flags; T.exc_off = c->d_exc_off; T.lut_h2s = c->d_lut_h2s; T.lut_s2h = c->d_lut_s2h; T.lut_okey = c->d_lut_okey; T.ovf = c->d_ovf;
    T.lut_oj = c->d_lut_oj; T.lut_omap = c->d_lut_omap; T.lut_ovi = c->d_lut_ovi; T.osave = c->d_osave;
    T.slots = c->d_slots; T.lut_slot = c->d_lut_slot; T.lut_h2c = c->d_lut_h2c; T.lut_h2c4 = c->d_lut_h2c4; T.zero_pad = c->d_zero; T.header_bytes = c->header_bytes;
    return T;
}

int ensure_scratch(snk_ctx_impl *c, size_t n_jobs, size_t n_out)
{
    if (n_jobs > c->jobs_cap) {
        dfree(c->d_jobs);
        HIPCHK(c, hipMalloc((void **)&c->d_jobs, n_jobs * sizeof(SnkJob)));
        c->jobs_cap = n_jobs;
    }
    if (n_out > c->out_cap) {
        dfree(c->d_out);
        HIPCHK(c, hipMalloc((void **)&c->d_out, n_out * sizeof(uint32_t)));
        c->out_cap = n_out;
    }
    return SNK_OK;
}

// Chains per wave of a 2-bit kernel workgroup under the current options (fast_lanes = 0: as many as the LDS holds).
int fast_geometry(snk_ctx_impl *c, uint32_t *lanes_out, uint32_t *short_last = nullptr)
{
    const uint32_t waves = (uint32_t)c->fast_waves;
    uint32_t lanes = (uint32_t)c->fast_lanes;
    // (a resident set with letters of the other case -- soft-masked stretches -- runs with the other case's LUT in LDS too: 2 KiB
    // fewer for the chains, 83 instead of 84: the last wave of a workgroup then runs one chain fewer, `short_last`)
    const size_t lut_b = c->any_other ? 2u * (size_t)SNK_FLUT_B : (size_t)SNK_FLUT_B;
    uint32_t shortl = 0u;
    if (lanes == 0u) {
        lanes = (uint32_t)((160 * 1024 - (size_t)SNK_FLUT_B) / ((size_t)waves * SNK_FCHAIN_B));
        if (lanes > 64u) lanes = 64u;
        if (lut_b + (size_t)lanes * waves * SNK_FCHAIN_B > 160 * 1024 &&
            !(lanes > 1u && lut_b + ((size_t)lanes * waves - 1u) * SNK_FCHAIN_B <= 160 * 1024))
            lanes = (uint32_t)((160 * 1024 - lut_b) / ((size_t)waves * SNK_FCHAIN_B));
    }
    // (one chain too many for the LDS -- 4 x 21 beside two LUTs: the last wave runs one fewer)
    if (lanes > 1u && lut_b + (size_t)lanes * waves * SNK_FCHAIN_B > 160 * 1024 && lut_b + ((size_t)lanes * waves - 1u) * SNK_FCHAIN_B <= 160 * 1024)
        shortl = 1u;
    if (lanes == 0u || lut_b + ((size_t)lanes * waves - shortl) * SNK_FCHAIN_B > 160 * 1024)
        return fail(c, SNK_E_ARG, "fast_lanes*fast_waves = %u chains exceed the 160 KiB LDS (max %d)", lanes * waves, c->any_other ? 83 : 84);
    *lanes_out = lanes;
    if (short_last) *short_last = shortl;
    return SNK_OK;
}

// Launch the kernels over a job list laid out as [2-bit jobs | linked byte jobs | one-shot jobs].
// tile: when non-NULL and its rows > 0, the 2-bit jobs are the dense tile (no list on the device).
struct SnkTileDesc { uint32_t r0 = 0, rows = 0, n = 0; bool ragged = false; };

int launch_jobs(snk_ctx_impl *c, hipStream_t st, const SnkJob *d_jobs, size_t n_fast, size_t n_bytes, size_t n_gen,
                uint32_t *d_out, bool singles = false, const SnkTileDesc *tile = nullptr)
{
    SnkTables T = make_tables(c);
    // (round 4) A set in which only SOME sequences carry exceptions: the pairs of two clean sequences -- the first n_clean 2-bit jobs of
    // the list, build_jobs puts them there -- run on the pure kernel (an exception kernel's ordinary loop exit costs four times the
    // pure kernel's); same geometry, same tables, one launch after the other on the stream.
    const size_t n_clean = (!singles && c->any_exc && !(tile && tile->rows > 0)) ? std::min(c->n_fast_clean, n_fast) : 0;      // (build_jobs decides)
    // (round 4) A dense tile of a set with exceptions: its rows go out in whole workgroups' worth.  The lanes of a wave serve each
    // other at the sites of the suffix they share, and a workgroup takes its chains' worth of consecutive jobs: with 1024 rows per
    // suffix (the full matrix in one launch) the workgroups straddle the suffixes and a wave's lanes stop sharing theirs -- ten runs
    // of N per Mbp 443 k pair-compr./s where 84 rows per launch give 512 k, 100 IUPAC codes 341 against 430 k, 5 % lower case 274
    // against 309 k.  So such a tile runs as the largest multiple of the workgroup's chains, then the rest (a launch that does not
    // fill the card: 2 % of a 1024-row matrix).  Pure sets gain a little too (the L1 serves a wave one suffix): 579 -> 584 k.
    uint32_t rows_a = (tile && tile->rows > 0) ? tile->rows : 0u, rows_b = 0u;
    if (rows_a && !singles && n_fast && (c->any_exc || (c->fast_spec <= 1 && c->far_lanes == 0))) {
        uint32_t gl = 0, gs = 0;
        if (fast_geometry(c, &gl, &gs) != SNK_OK) return SNK_E_ARG;
        const uint32_t cwg = gl * (uint32_t)c->fast_waves - gs;
        if (cwg && rows_a > cwg && rows_a % cwg) { rows_b = rows_a % cwg; rows_a -= rows_b; }
    }
    uint32_t *const d_out_all = d_out;
    for (int grp = 0; grp < 2; ++grp) {
        const size_t nf_g = rows_b ? (size_t)(grp == 0 ? rows_a : rows_b) * tile->n
                                   : grp == 0 ? (n_clean ? n_clean : n_fast) : (n_clean ? n_fast - n_clean : 0);
        if (!nf_g) continue;
        uint32_t *const d_out = d_out_all + (rows_b && grp == 1 ? (size_t)rows_a * tile->n : 0);      // (this part's rows)
        const SnkJob *const jl_g = grp == 0 ? d_jobs : d_jobs + n_clean;
        uint32_t waves = (uint32_t)c->fast_waves;
        const bool exc = c->any_exc && !(n_clean && grp == 0);      // some resident 2-bit sequence has exceptions: the instantiations that know about them
        uint32_t lanes = 0, short_last = 0;
        if (fast_geometry(c, &lanes, &short_last) != SNK_OK) return SNK_E_ARG;
        if (singles && c->fast_lanes == 0) {
            short_last = 0u;
            // Phase A has N jobs, not N^2, and no two of them share a byte: what counts is one chain's serial parse.  Measured
            // (tools/gpu_singles.py, tools/gpu_geom.py, 1024 x 1 Mbp): 84-chain workgroups on 13 CUs 85 ms (one lane per chain) /
            // 62 ms (two); spread over all CUs as 4 waves x 1 chain 92 ms -- waves with fewer than ~8 chains slow each other
            // down when several share a CU (1 / 2 / 3 / 4 waves of 1 chain: 44 / 60 / 76 / 87 ms per round, waiting for
            // instruction issue by the SQ counters; 8 chains per wave and more: no such effect) -- and as ONE wave per CU with
            // ceil(N / CUs) chains 48 ms.  So: one wave per workgroup while 21 chains per CU cover the jobs, more waves beyond.
            const uint64_t cus = (uint64_t)std::max(c->n_cus, 1);
            const uint32_t full = lanes;                                  // chains per wave the LDS allows at c->fast_waves
            waves = (uint32_t)std::min<uint64_t>(waves, std::max<uint64_t>(1u, (nf_g + cus * full - 1u) / (cus * full)));
            lanes = std::min<uint32_t>(full, (uint32_t)std::max<uint64_t>(1u, (nf_g + cus * waves - 1u) / (cus * waves)));
        }
        const bool tri = (c->fast_spec == 3 || c->fast_spec == 36) && !c->any_exc && !singles && c->far_lanes == 0;     // three lanes per chain: 5 chains per 16-lane row
        if (tri && lanes > 20u) lanes = 20u;
        const uint32_t chains = lanes * waves;
        const size_t lds = (c->any_other ? 2u : 1u) * (size_t)SNK_FLUT_B + (size_t)(chains - short_last) * SNK_FCHAIN_B;
        SnkFastGrid G;
        G.short_last = short_last; G.flut = (c->any_other ? 2u : 1u) * SNK_FLUT_B;
        const bool dense = tile && tile->rows > 0;
        // far chains (tables in global memory, extra waves): pair launches of pure-ACGT sets that fill the card
        uint32_t far_waves = (!c->any_exc && !singles && c->far_lanes > 0) ? (uint32_t)c->far_waves : 0u;
        if (far_waves && (waves + far_waves > 8u || nf_g < (size_t)c->far_min * chains * (size_t)std::max(c->n_cus, 1))) far_waves = 0u;
        G.jobs = dense ? nullptr : jl_g; G.n_jobs = (uint32_t)nf_g;
        G.r0 = dense ? tile->r0 + (rows_b && grp == 1 ? rows_a : 0u) : 0u;
        G.rows = dense ? (rows_b ? (grp == 0 ? rows_a : rows_b) : tile->rows) : 1u; G.n = dense ? tile->n : 1u;
        G.batch = lanes; G.queue = nullptr; G.yorder = nullptr;
        const bool spec = c->fast_spec != 0 && far_waves == 0u && lanes <= 32u;       // two lanes per chain
        const void *fk = singles ? (exc ? (spec ? (const void *)snk_fastx_singles_kernel : (const void *)snk_fastx_singles_one_kernel)
                                        : (spec ? (const void *)snk_fast_singles_kernel : (const void *)snk_fast_singles_one_kernel))
                       : exc ? (spec ? (c->fast_asm ? (const void *)snk_fastx_kernel : (const void *)snk_fastx_spec_cxx_kernel)
                                     : (c->fast_asm ? (const void *)snk_fastx_one_kernel : (const void *)snk_fastx_cxx_kernel))
                             : tri ? (c->fast_spec == 36 ? (const void *)snk_fast_tri6_cxx_kernel : (const void *)snk_fast_tri_cxx_kernel)
                             : (spec ? (c->fast_asm ? (const void *)snk_fast_kernel : (const void *)snk_fast_spec_cxx_kernel)
                                     : (c->fast_asm ? (const void *)snk_fast_one_kernel : (const void *)snk_fast_cxx_kernel));
        HIPCHK(c, hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        {   // the kernel addresses its slot LUT at LDS offset 0: that holds only without static LDS
            hipFuncAttributes fa;
            HIPCHK(c, hipFuncGetAttributes(&fa, fk));
            if (fa.sharedSizeBytes != 0)
                return fail(c, SNK_E_STATE, "snk_fast_kernel has %zu bytes of static LDS", (size_t)fa.sharedSizeBytes);
        }
        // persistent launch: one workgroup per compute unit at most (the LDS admits one), waves walk the batches
        uint32_t grid = (uint32_t)((nf_g + chains - 1) / chains);
        if (c->n_cus > 0 && grid > (uint32_t)c->n_cus) grid = (uint32_t)c->n_cus;
        // (far waves are slower per chain than LDS waves: only the atomic queue keeps both kinds busy)
        const bool dynamic = far_waves ? true : c->fast_dynamic < 0 ? (singles || (tile && tile->ragged)) : c->fast_dynamic != 0;
        if (dynamic) {       // ragged lengths: later batches come from an atomic counter that starts behind the waves' first ones
            if (dense) G.yorder = c->d_yorder;               // ... and the longest suffixes go out first
            G.queue = c->d_queue + (c->queue_next++ & 63u);
            HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)G.queue, 0, 1, st));
        }
        if (far_waves) {
            const size_t need = (size_t)grid * far_waves * (size_t)c->far_lanes * SNK_FSLOTS * sizeof(uint32_t);
            if (need > c->far_bytes) {
                HIPCHK(c, hipStreamSynchronize(st));
                if (c->far_in_flight) { HIPCHK(c, hipEventSynchronize(c->far_busy)); c->far_in_flight = false; }
                dfree(c->d_far);
                HIPCHK(c, hipMalloc((void **)&c->d_far, need));
                c->far_bytes = need;
            }
            // one set of far tables per context: launches that use it run one after the other, whatever their streams
            if (c->far_in_flight) HIPCHK(c, hipStreamWaitEvent(st, c->far_busy, 0));
            G.far_tab = c->d_far; G.lds_waves = waves; G.far_lanes = (uint32_t)c->far_lanes;
            // a far chain takes ~1.4x as long over a job as an LDS chain: below ~1.4 jobs per LDS chain left, the LDS waves finish first
            G.far_stop = (uint32_t)(((uint64_t)grid * chains * (uint64_t)c->far_stop_pct) / 100u);
        }
        if (exc) {
            const size_t need = (size_t)grid * chains * 4096u * sizeof(uint32_t);       // one overflow table per resident chain
            if (need > c->ovf_bytes) {
                HIPCHK(c, hipStreamSynchronize(st));
                if (c->ovf_in_flight) { HIPCHK(c, hipEventSynchronize(c->ovf_busy)); c->ovf_in_flight = false; }
                dfree(c->d_ovf); dfree(c->d_osave);
                HIPCHK(c, hipMalloc((void **)&c->d_ovf, need));
                HIPCHK(c, hipMalloc((void **)&c->d_osave, need / 8u));      // 2 KiB per chain: its set-case table while the other case's is in LDS
                c->ovf_bytes = need;
                T.ovf = c->d_ovf; T.osave = c->d_osave;
            }
        }
        // (launches with exceptions share the context's overflow tables: one at a time, whatever their streams)
        if (exc && c->ovf_in_flight) HIPCHK(c, hipStreamWaitEvent(st, c->ovf_busy, 0));
        {
            typedef void (*SnkFastKernel)(SnkTables, SnkFastGrid, uint32_t, uint32_t *, uint32_t *);
            hipLaunchKernelGGL((SnkFastKernel)fk, dim3(grid), dim3(64 * (waves + far_waves)), lds, st, T, G, lanes, d_out, c->d_status);
        }
        HIPCHK(c, hipGetLastError());
        if (far_waves) { HIPCHK(c, hipEventRecord(c->far_busy, st)); c->far_in_flight = true; }
        if (exc) { HIPCHK(c, hipEventRecord(c->ovf_busy, st)); c->ovf_in_flight = true; }
    }
    // tables in global memory ("bytes_gt"): by default for the full table (measured: 1.9 x on ACGT text, 3.3 x on protein;
    // the compact tables are faster in LDS), once the jobs would take the LDS kernel more than two rounds of the card
    int gt_waves = c->bytes_gt;
    if (gt_waves < 0)
        gt_waves = (!c->compact_ok && n_bytes >= 2u * (size_t)(c->bytes_lanes * c->bytes_waves) * (size_t)std::max(c->n_cus, 1)) ? 4 : 0;
    if (singles || c->bytes_legacy || (c->compact_ok && !c->d_slots)) gt_waves = 0;
    // (both kernels at once on two streams, sharing the jobs -- the global-table kernel needs no LDS -- was measured:
    // 119-126 k pair-compr/s on the compact set against 132 k for the LDS kernel alone)
    const size_t n_gt = gt_waves > 0 ? n_bytes : 0;
    const size_t n_lds = n_bytes - n_gt;
    if (n_gt) {
        // every lane runs a chain; one launch per `cap` jobs (one table each)
        const int capn = c->compact_ok ? c->compact_cap : 0;
        const uint32_t ns = capn == 2048 ? SnkGT<2048>::NS : capn == 1024 ? SnkGT<1024>::NS : SnkGT<0>::NS;
        const uint32_t threads = 64u * (uint32_t)gt_waves;
        const size_t cap = (size_t)std::max(c->n_cus, 1) * (size_t)c->bytes_gt_wgs * threads;
        const size_t need = std::min(cap, n_gt) * ns * sizeof(uint32_t);
        if (need > c->bgt_bytes) {
            if (c->bgt_in_flight) { HIPCHK(c, hipEventSynchronize(c->bgt_busy)); c->bgt_in_flight = false; }
            dfree(c->d_bgt); c->bgt_bytes = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_bgt, need));      // (fine-grained: the same rate; uncached: 20 % slower)
            c->bgt_bytes = need;
        }
        hipStream_t gst = st;
        if (c->bgt_in_flight) HIPCHK(c, hipStreamWaitEvent(gst, c->bgt_busy, 0));
        for (size_t done = 0; done < n_gt; done += cap) {
            const uint32_t nj = (uint32_t)std::min(cap, n_gt - done);
            const uint32_t grid = (nj + threads - 1u) / threads;
            const SnkJob *jb = d_jobs + n_fast + n_lds + done;
            if (capn == 2048)      hipLaunchKernelGGL(snk_bytes_gt2k_kernel, dim3(grid), dim3(threads), 0, gst, T, jb, nj, c->d_bgt, d_out, c->d_status);
            else if (capn == 1024) hipLaunchKernelGGL(snk_bytes_gt1k_kernel, dim3(grid), dim3(threads), 0, gst, T, jb, nj, c->d_bgt, d_out, c->d_status);
            else                   hipLaunchKernelGGL(snk_bytes_gt_kernel,   dim3(grid), dim3(threads), 0, gst, T, jb, nj, c->d_bgt, d_out, c->d_status);
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipEventRecord(c->bgt_busy, gst)); c->bgt_in_flight = true;
    }
    const size_t n_bytes_all = n_bytes;
    n_bytes = n_lds;
    if (n_bytes && c->compact_ok) {
        const bool big = c->compact_cap == 2048;
        const uint32_t lanes = (uint32_t)(big ? c->c2bytes_lanes : c->cbytes_lanes);
        const uint32_t waves = (uint32_t)(big ? c->c2bytes_waves : c->cbytes_waves);
        const uint32_t chains = lanes * waves;
        const size_t chain_b = big ? SnkBT<2048, false>::CHAIN_B : SnkBT<1024, false>::CHAIN_B;
        const size_t lds = (size_t)SnkBT<1024, false>::LUT_B + (size_t)chains * chain_b;
        const bool spec = c->bytes_spec && c->d_slots && lanes <= 32u;       // two lanes per chain
        const void *kern = spec ? (big ? (const void *)snk_bytes_compact2k_spec_kernel : (const void *)snk_bytes_compact_spec_kernel)
                                : (big ? (const void *)snk_bytes_compact2k_kernel : (const void *)snk_bytes_compact_kernel);
        if (lds > 160 * 1024)
            return fail(c, SNK_E_ARG, "%u compact byte chains exceed the 160 KiB LDS (max %d)", chains, big ? 35 : 70);
        HIPCHK(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        {   // the kernel addresses its hash LUT at LDS offset 0: that holds only without static LDS
            hipFuncAttributes fa;
            HIPCHK(c, hipFuncGetAttributes(&fa, kern));
            if (fa.sharedSizeBytes != 0)
                return fail(c, SNK_E_STATE, "compact byte kernel has %zu bytes of static LDS", (size_t)fa.sharedSizeBytes);
        }
        const uint32_t grid = (uint32_t)((n_bytes + chains - 1) / chains);
        {
            typedef void (*SnkBytesKernel)(SnkTables, const SnkJob *, uint32_t, uint32_t, uint32_t *, uint32_t *);
            hipLaunchKernelGGL((SnkBytesKernel)kern, dim3(grid), dim3(64 * waves), lds, st,
                               T, (const SnkJob *)(d_jobs + n_fast), (uint32_t)n_bytes, lanes, d_out, c->d_status);
        }
        HIPCHK(c, hipGetLastError());
    } else if (n_bytes) {
        const uint32_t lanes = (uint32_t)c->bytes_lanes, waves = (uint32_t)c->bytes_waves;
        const uint32_t chains = lanes * waves;
        const size_t lds = (size_t)chains * SnkBT<0, false>::CHAIN_B;
        if (lds > 160 * 1024)
            return fail(c, SNK_E_ARG, "bytes_lanes*bytes_waves = %u chains exceed the 160 KiB LDS (max 18)", chains);
        const bool spec = c->bytes_spec && c->d_slots && lanes <= 32u;
        const void *kern = spec ? (const void *)snk_bytes_spec_kernel : (const void *)snk_bytes_kernel;
        HIPCHK(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint32_t grid = (uint32_t)((n_bytes + chains - 1) / chains);
        {
            typedef void (*SnkBytesKernel)(SnkTables, const SnkJob *, uint32_t, uint32_t, uint32_t *, uint32_t *);
            hipLaunchKernelGGL((SnkBytesKernel)kern, dim3(grid), dim3(64 * waves), lds, st,
                               T, (const SnkJob *)(d_jobs + n_fast), (uint32_t)n_bytes, lanes, d_out, c->d_status);
        }
        HIPCHK(c, hipGetLastError());
    }
    n_bytes = n_bytes_all;
    if (n_gen && !c->bytes_legacy) {
        // one-shot inputs (n <= 64 KiB): the tight-loop kernel in one-shot mode
        const bool cmp = c->oneshot_compact_ok;
        const uint32_t lanes = (uint32_t)(cmp ? c->cos_lanes : c->os_lanes);
        const uint32_t waves = (uint32_t)(cmp ? c->cos_waves : c->os_waves);
        const uint32_t chains = lanes * waves;
        const size_t lds = cmp ? (size_t)SnkBT<1024, true>::LUT_B + (size_t)chains * SnkBT<1024, true>::CHAIN_B
                               : (size_t)chains * SnkBT<0, true>::CHAIN_B;
        const void *kern = cmp ? (const void *)snk_oneshot_compact_kernel : (const void *)snk_oneshot_kernel;
        if (lds > 160 * 1024)
            return fail(c, SNK_E_ARG, "%u one-shot chains exceed the 160 KiB LDS (max %d)", chains, cmp ? 67 : 9);
        HIPCHK(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (cmp) {
            hipFuncAttributes fa;
            HIPCHK(c, hipFuncGetAttributes(&fa, kern));
            if (fa.sharedSizeBytes != 0)
                return fail(c, SNK_E_STATE, "compact one-shot kernel has %zu bytes of static LDS", (size_t)fa.sharedSizeBytes);
        }
        const uint32_t grid = (uint32_t)((n_gen + chains - 1) / chains);
        if (cmp)
            hipLaunchKernelGGL(snk_oneshot_compact_kernel, dim3(grid), dim3(64 * waves), lds, st,
                               T, d_jobs + n_fast + n_bytes, (uint32_t)n_gen, lanes, d_out, c->d_status);
        else
            hipLaunchKernelGGL(snk_oneshot_kernel, dim3(grid), dim3(64 * waves), lds, st,
                               T, d_jobs + n_fast + n_bytes, (uint32_t)n_gen, lanes, d_out, c->d_status);
        HIPCHK(c, hipGetLastError());
    } else if (n_gen) {
        const uint32_t chains = (uint32_t)c->gen_chains;
        const size_t lds = (size_t)chains * 16384;
        HIPCHK(c, hipFuncSetAttribute((const void *)snk_generic_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint32_t grid = (uint32_t)((n_gen + chains - 1) / chains);
        hipLaunchKernelGGL(snk_generic_kernel, dim3(grid), dim3(64), lds, st,
                           T, d_jobs + n_fast + n_bytes, (uint32_t)n_gen, chains, d_out, c->d_status);
        HIPCHK(c, hipGetLastError());
    }
    return SNK_OK;
}

int check_status(snk_ctx_impl *c)
{
    uint32_t st = 0;
    HIPCHK(c, hipMemcpy(&st, c->d_status, sizeof st, hipMemcpyDeviceToHost));
    if (st) {
        (void)hipMemset(c->d_status, 0, sizeof(uint32_t));
        return fail(c, SNK_E_KERNEL, "device-side check failed (status 0x%x)", st);
    }
    return SNK_OK;
}

bool pair_is_fast(const snk_ctx_impl *c, int i, int j)
{
    return !c->force_generic && c->is_packed[i] && c->is_packed[j] &&
           (uint64_t)c->len[i] + c->len[j] > SNK_BLOCK;
}

// Two launches leave the card part-empty twice at their ends (a round of jobs too many in the worst case): measured at 1024 x 1 Mbp
// with ten runs of N in 10 % of the genomes, the full matrix (49 rounds of the card) + 7.6 %, 336 rows (16 rounds) - 1.8 %, 84 rows
// (4 rounds) - 18 %.  So a list is split (option split_clean = 1) only when its clean part alone fills the card 16 times.
size_t split_threshold(const snk_ctx_impl *c) { return (size_t)16 * 84u * (size_t)std::max(c->n_cus, 1); }

// Build the job list for an arbitrary set of ordered pairs (i, j) -> out index.
// Fast jobs are ordered by suffix sequence j so that the chains of one workgroup
// walk the same bytes (L1/L2 locality); generic jobs follow.
template <typename PairAt>
int build_jobs(snk_ctx_impl *c, size_t n_pairs, PairAt pair_at, size_t &n_fast, size_t &n_bytes, size_t &n_gen)
{
    std::vector<SnkJob> fast, bytes, gen;
    fast.reserve(n_pairs);
    for (size_t t = 0; t < n_pairs; ++t) {
        int i, j; uint32_t o;
        pair_at(t, i, j, o);
        if (i < 0 || j < 0 || i >= c->n || j >= c->n) return fail(c, SNK_E_ARG, "pair index out of range");
        const uint64_t n = (uint64_t)c->len[i] + c->len[j];
        if (n >= 0x7E000000ull) return fail(c, SNK_E_TOOBIG, "concatenation of %d and %d too long", i, j);
        SnkJob jb; jb.xi = i; jb.yi = j; jb.out_idx = o; jb.snap = 0;
        if (pair_is_fast(c, i, j)) fast.push_back(jb);
        else if (n > SNK_BLOCK && !c->bytes_legacy) bytes.push_back(jb);
        else gen.push_back(jb);
    }
    n_fast = fast.size(); n_bytes = bytes.size(); n_gen = gen.size();
    c->n_fast_clean = 0;
    if (c->split_clean && c->any_exc && !fast.empty()) {
        // pairs of two sequences without exceptions first (each part keeps its suffix-major order) -- when launch_jobs will split the
        // list (see there); a list that stays whole keeps its order: its waves' chains share their suffixes
        const auto clean = [&](const SnkJob &jb) { return !c->has_exc[(size_t)jb.xi] && !c->has_exc[(size_t)jb.yi]; };
        const size_t n_cl = (size_t)std::count_if(fast.begin(), fast.end(), clean);
        if (n_cl == fast.size()) c->n_fast_clean = n_cl;
        else if (n_cl && (c->split_clean == 2 || n_cl >= split_threshold(c))) {
            std::stable_partition(fast.begin(), fast.end(), clean);
            c->n_fast_clean = n_cl;
        }
    }
    c->dense_tile = n_fast == n_pairs && c->n_fast_clean == 0;
    c->h_jobs.swap(fast);
    c->h_jobs.insert(c->h_jobs.end(), bytes.begin(), bytes.end());
    c->h_jobs.insert(c->h_jobs.end(), gen.begin(), gen.end());
    return SNK_OK;
}

// One launch over the job list in c->h_jobs (or over a dense tile: nothing is copied then).  Every launch gets
// its own hipEvent pair (snk_last_pairs_ms, snk_pairs_ms_log).  The device copy of the list is ONE buffer per
// context: a launch that needs it waits (on the host) for the previous launch that read it, on whatever stream
// that one ran, so interleaved calls on two streams stay correct; dense tiles share nothing.
int run_pairs(snk_ctx_impl *c, hipStream_t st, size_t n_fast, size_t n_bytes, size_t n_gen, uint32_t *d_out,
              const SnkTileDesc *tile = nullptr)
{
    const size_t nj = n_fast + n_bytes + n_gen;
    if (nj == 0) return SNK_OK;
    const bool dense = tile && tile->rows > 0;
    if (!dense) {
        if (c->jobs_in_flight) { HIPCHK(c, hipEventSynchronize(c->jobs_busy)); c->jobs_in_flight = false; }
        HIPCHK(c, hipMemcpyAsync(c->d_jobs, c->h_jobs.data(), nj * sizeof(SnkJob), hipMemcpyHostToDevice, st));
    }
    if (!c->ev_logging) c->ev_used = 0;          // nobody reads the log: slot 0 serves every launch (snk_last_pairs_ms)
    if (c->ev_used == c->ev_log.size()) {
        if (c->ev_log.size() < 4096) {
            hipEvent_t a = nullptr, b = nullptr;
            HIPCHK(c, hipEventCreate(&a));
            HIPCHK(c, hipEventCreate(&b));
            c->ev_log.emplace_back(a, b);
        } else {
            c->ev_used = c->ev_log.size() - 1;      // log full: the last slot is reused (snk_pairs_ms_log reports what it has)
        }
    }
    c->ev0 = c->ev_log[c->ev_used].first; c->ev1 = c->ev_log[c->ev_used].second; c->ev_used++;
    HIPCHK(c, hipEventRecord(c->ev0, st));
    int rc = launch_jobs(c, st, c->d_jobs, n_fast, n_bytes, n_gen, d_out, false, tile);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, st));
    if (!dense) { HIPCHK(c, hipEventRecord(c->jobs_busy, st)); c->jobs_in_flight = true; }
    c->ev_valid = true;
    c->ms_accum = -1.0;
    return SNK_OK;
}

// ---- phase A (ref:snacc/cli.py:108-116): single sizes + prefix snapshots of the sequences `rows` ----------------
// Blocking (the snapshots must be complete before a pair launch on any stream starts from them).
int run_singles(snk_ctx_impl *c, const std::vector<uint32_t> &rows)
{
    if (rows.empty()) return SNK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->jobs_in_flight) { HIPCHK(c, hipEventSynchronize(c->jobs_busy)); c->jobs_in_flight = false; }
    std::vector<SnkJob> fast, bytes, gen;
    std::vector<uint32_t> conv;
    for (const uint32_t g : rows) {
        const uint32_t spos = c->len[g] > SNK_BLOCK ? c->len[g] / SNK_BLOCK * SNK_BLOCK : 0u;
        SnkJob jb; jb.xi = (int)g; jb.yi = -1; jb.out_idx = g; jb.snap = spos ? 1 : 0;
        const bool f = !c->force_generic && c->is_packed[g] && c->len[g] > SNK_BLOCK;
        if (f) { fast.push_back(jb); if (!c->any_exc) conv.push_back(g); }     // (the fastx kernel dumps snap_gen itself)
        else if (c->len[g] > SNK_BLOCK && !c->bytes_legacy) bytes.push_back(jb);
        else gen.push_back(jb);
    }
    const size_t nf = fast.size(), nb = bytes.size(), ng = gen.size(), nj = nf + nb + ng;
    c->h_jobs = fast;
    c->h_jobs.insert(c->h_jobs.end(), bytes.begin(), bytes.end());
    c->h_jobs.insert(c->h_jobs.end(), gen.begin(), gen.end());
    int rc = ensure_scratch(c, nj, 0);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, c->h_jobs.data(), nj * sizeof(SnkJob), hipMemcpyHostToDevice, c->stream));
    rc = launch_jobs(c, c->stream, c->d_jobs, nf, nb, ng, c->d_single, true);
    if (rc) return rc;
    DevTemp<uint32_t> ids;
    if (!conv.empty()) {
        HIPCHK(c, hipMalloc((void **)&ids.p, conv.size() * 4));
        HIPCHK(c, hipMemcpyAsync(ids.p, conv.data(), conv.size() * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(snk_snap_convert_kernel, dim3((uint32_t)conv.size()), dim3(256), 0, c->stream,
                           c->d_snap_fast, c->d_snap_gen, c->d_lut_hash, c->d_lut_slot, ids.p, (uint32_t)conv.size());
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = check_status(c);
    if (rc) return rc;
    for (const uint32_t g : rows) c->single_have[g] = 1;
    return SNK_OK;
}

// phase A for the sequences [r0, r1) that have not had it yet
int ensure_singles(snk_ctx_impl *c, int r0, int r1)
{
    std::vector<uint32_t> rows;
    for (int g = r0; g < r1; ++g)
        if (!c->single_have[(size_t)g]) rows.push_back((uint32_t)g);
    return run_singles(c, rows);
}

} // namespace

struct snk_ctx : snk_ctx_impl {};

// ---- narrow view of the context for the deflate translation unit (snk_internal.h) ------------
int snk_internal_view(snk_ctx *c, SnkSeqView *v)
{
    if (!c || !v) return SNK_E_ARG;
    v->device = c->device; v->stream = c->stream; v->n = c->n;
    v->len = c->len.data(); v->boff = c->boff.data(); v->d_bytes = c->d_bytes;
    v->dfl_serial = c->dfl_serial ? 1 : 0;
    v->dfl_kmer = c->dfl_kmer ? 1 : 0;
    v->dfl_norestart = c->dfl_norestart ? 1 : 0;
    return SNK_OK;
}
int snk_internal_fail(snk_ctx *c, int code, const char *msg) { return fail(c, code, "%s", msg); }
void **snk_internal_dfl_slot(snk_ctx *c, void (***free_fn)(void *))
{
    if (free_fn) *free_fn = &c->dfl_free;
    return &c->dfl;
}

extern "C" {

int snk_version(void) { return SNK_ABI_VERSION; }

const char *snk_last_error(const snk_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int snk_ctx_create(int device, snk_ctx **out)
{
    if (!out) return fail(nullptr, SNK_E_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SNK_E_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, SNK_E_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    snk_ctx *c = new (std::nothrow) snk_ctx();
    if (!c) return fail(nullptr, SNK_E_HIP, "out of host memory");
    c->device = device;
#define CRCHK(call)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            fail(nullptr, SNK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));             \
            snk_ctx_destroy(c);                                                                  \
            return SNK_E_HIP;                                                                    \
        }                                                                                        \
    } while (0)
    CRCHK(hipSetDevice(device));
    CRCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CRCHK(hipEventCreate(&c->jobs_busy));
    CRCHK(hipEventCreateWithFlags(&c->ovf_busy, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&c->far_busy, hipEventDisableTiming));
    CRCHK(hipEventCreateWithFlags(&c->bgt_busy, hipEventDisableTiming));
    {
        hipDeviceProp_t prop;
        CRCHK(hipGetDeviceProperties(&prop, device));
        c->n_cus = prop.multiProcessorCount;
    }
    CRCHK(hipMalloc((void **)&c->d_queue, 64 * sizeof(uint32_t)));
    CRCHK(hipMemset(c->d_queue, 0, 64 * sizeof(uint32_t)));
    CRCHK(hipMalloc((void **)&c->d_status, sizeof(uint32_t)));
    CRCHK(hipMemset(c->d_status, 0, sizeof(uint32_t)));
    CRCHK(hipMalloc((void **)&c->d_zero, 4 * SNK_PAD));
    CRCHK(hipMemset(c->d_zero, 0, 4 * SNK_PAD));
    {
        CRCHK(hipMalloc((void **)&c->d_lut_h2c, 4096 * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_h2c4, 8192 * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_hashset, (128 + 256) * sizeof(uint32_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_slot, 1024 * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_hash, 1024 * sizeof(uint32_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_h2s, 4096 * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_s2h, SNK_FSLOTS * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_okey, 1024 * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_oj, 1024 * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_ovi, 4096 * sizeof(uint16_t)));
        CRCHK(hipMalloc((void **)&c->d_lut_omap, SNK_FSLOTS * sizeof(uint32_t)));
        if (upload_luts(c) != SNK_OK) { snk_ctx_destroy(c); return SNK_E_STATE; }
    }
#undef CRCHK
    *out = c;
    return SNK_OK;
}

void snk_ctx_destroy(snk_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_sequences(c);
    dfree(c->d_lut_h2s); dfree(c->d_lut_s2h); dfree(c->d_lut_okey); dfree(c->d_lut_oj); dfree(c->d_lut_ovi); dfree(c->d_lut_omap);
    dfree(c->d_zero); dfree(c->d_lut_slot); dfree(c->d_lut_hash); dfree(c->d_lut_h2c); dfree(c->d_lut_h2c4); dfree(c->d_hashset); dfree(c->d_status);
    dfree(c->d_jobs); dfree(c->d_out);
    for (auto &e : c->ev_log) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->jobs_busy) (void)hipEventDestroy(c->jobs_busy);
    if (c->ovf_busy) (void)hipEventDestroy(c->ovf_busy);
    if (c->far_busy) (void)hipEventDestroy(c->far_busy);
    if (c->bgt_busy) (void)hipEventDestroy(c->bgt_busy);
    dfree(c->d_queue);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int snk_set_option(snk_ctx *c, const char *key, long value)
{
    if (!c || !key) return fail(c, SNK_E_ARG, "NULL argument");
    std::string k(key);
    if (k == "fast_lanes") {
        if (value < 0 || value > 64) return fail(c, SNK_E_ARG, "fast_lanes must be 0 (as many as fit) or 1..64");
        c->fast_lanes = (int)value;
    } else if (k == "fast_waves") {
        if (value < 1 || value > 8) return fail(c, SNK_E_ARG, "fast_waves must be 1..8");
        c->fast_waves = (int)value;
    } else if (k == "far_lanes") {
        if (value < 0 || value > 64) return fail(c, SNK_E_ARG, "far_lanes must be 0 (no far chains) or 1..64");
        c->far_lanes = (int)value;
    } else if (k == "far_waves") {
        if (value < 1 || value > 7) return fail(c, SNK_E_ARG, "far_waves must be 1..7");
        c->far_waves = (int)value;
    } else if (k == "far_min") {
        if (value < 0 || value > 1000) return fail(c, SNK_E_ARG, "far_min must be 0..1000");
        c->far_min = (int)value;
    } else if (k == "far_stop_pct") {
        if (value < 0 || value > 10000) return fail(c, SNK_E_ARG, "far_stop_pct must be 0..10000");
        c->far_stop_pct = (int)value;
    } else if (k == "fast_spec") {
        if (value != 0 && value != 1 && value != 3 && value != 36) return fail(c, SNK_E_ARG, "fast_spec must be 0, 1, 3 or 36");
        c->fast_spec = (int)value;
    } else if (k == "defer_singles") {
        c->defer_singles = value != 0;
    } else if (k == "split_clean") {
        if (value < 0 || value > 2) return fail(c, SNK_E_ARG, "split_clean must be 0, 1 or 2");
        c->split_clean = (int)value;
    } else if (k == "fast_asm") {
        c->fast_asm = value != 0;
    } else if (k == "exc_limit") {
        if (c->n) return fail(c, SNK_E_STATE, "exc_limit must be set before snk_upload");
        if (value < 0 || value > 65536) return fail(c, SNK_E_ARG, "exc_limit must be 0..65536 granules per 2^20 bases");
        c->exc_limit = value;
    } else if (k == "arena_limit") {
        if (value < 4096 || (uint64_t)value > 0xFFF00000ull) return fail(c, SNK_E_ARG, "arena_limit must be 4096..0xFFF00000");
        c->arena_limit = (uint64_t)value;
    } else if (k == "fast_dynamic") {
        if (value < -1 || value > 1) return fail(c, SNK_E_ARG, "fast_dynamic must be -1 (auto), 0 or 1");
        c->fast_dynamic = (int)value;
    } else if (k == "gen_chains") {
        if (value < 1 || value > 9) return fail(c, SNK_E_ARG, "gen_chains must be 1..9");
        c->gen_chains = (int)value;
    } else if (k == "bytes_spec") {
        if (value < 0 || value > 1) return fail(c, SNK_E_ARG, "bytes_spec must be 0 or 1");
        c->bytes_spec = (int)value;
    } else if (k == "bytes_gt") {
        if (value < -1 || value > 8) return fail(c, SNK_E_ARG, "bytes_gt must be -1..8 (waves per workgroup; 0 = tables in LDS, -1 = choose)");
        c->bytes_gt = (int)value;
    } else if (k == "bytes_gt_wgs") {
        if (value < 1 || value > 16) return fail(c, SNK_E_ARG, "bytes_gt_wgs must be 1..16");
        c->bytes_gt_wgs = (int)value;
    } else if (k == "bytes_lanes") {
        if (value < 1 || value > 64) return fail(c, SNK_E_ARG, "bytes_lanes must be 1..64");
        c->bytes_lanes = (int)value;
    } else if (k == "bytes_waves") {
        if (value < 1 || value > 16) return fail(c, SNK_E_ARG, "bytes_waves must be 1..16");
        c->bytes_waves = (int)value;
    } else if (k == "cbytes_lanes") {
        if (value < 1 || value > 64) return fail(c, SNK_E_ARG, "cbytes_lanes must be 1..64");
        c->cbytes_lanes = (int)value;
    } else if (k == "cbytes_waves") {
        if (value < 1 || value > 16) return fail(c, SNK_E_ARG, "cbytes_waves must be 1..16");
        c->cbytes_waves = (int)value;
    } else if (k == "os_lanes" || k == "os_waves" || k == "cos_lanes" || k == "cos_waves") {
        if (value < 1 || value > 64) return fail(c, SNK_E_ARG, "%s must be 1..64", key);
        (k == "os_lanes" ? c->os_lanes : k == "os_waves" ? c->os_waves : k == "cos_lanes" ? c->cos_lanes : c->cos_waves) = (int)value;
    } else if (k == "c2bytes_lanes") {
        if (value < 1 || value > 64) return fail(c, SNK_E_ARG, "c2bytes_lanes must be 1..64");
        c->c2bytes_lanes = (int)value;
    } else if (k == "c2bytes_waves") {
        if (value < 1 || value > 16) return fail(c, SNK_E_ARG, "c2bytes_waves must be 1..16");
        c->c2bytes_waves = (int)value;
    } else if (k == "bytes_compact") {
        if (c->n) return fail(c, SNK_E_STATE, "bytes_compact must be set before snk_upload");
        c->bytes_compact_opt = (int)value;
    } else if (k == "bytes_legacy") {
        c->bytes_legacy = value != 0;
    } else if (k == "force_generic") {
        c->force_generic = value != 0;
    } else if (k == "deflate_serial") {
        c->dfl_serial = value != 0;
    } else if (k == "deflate_kmer") {
        c->dfl_kmer = value != 0;
    } else if (k == "deflate_norestart") {
        c->dfl_norestart = value != 0;
    } else if (k == "content_size") {
        if (c->n) return fail(c, SNK_E_STATE, "content_size must be set before snk_upload");
        c->header_bytes = value ? 15u : 7u;
    } else {
        return fail(c, SNK_E_ARG, "unknown option '%s'", key);
    }
    return SNK_OK;
}

int snk_num_sequences(const snk_ctx *c) { return c ? c->n : SNK_E_ARG; }
int snk_num_packed(const snk_ctx *c) { return c ? c->n_packed : SNK_E_ARG; }
int snk_fast_chains(snk_ctx *c)
{
    if (!c) return SNK_E_ARG;
    uint32_t lanes = 0, short_last = 0;
    if (fast_geometry(c, &lanes, &short_last) != SNK_OK) return SNK_E_ARG;
    return (int)(lanes * (uint32_t)c->fast_waves - short_last);
}
int snk_lengths(const snk_ctx *c, uint64_t *lens)
{
    if (!c || (c->n && !lens)) return SNK_E_ARG;
    for (int g = 0; g < c->n; ++g) lens[g] = c->len[(size_t)g];
    return SNK_OK;
}
int snk_num_compact_hashes(const snk_ctx *c) { return c ? (c->compact_ok ? c->n_hashes : 0) : SNK_E_ARG; }

static int upload_impl(snk_ctx *c, int n_seq, const uint8_t *const *seqs, const uint64_t *lens);

int snk_upload(snk_ctx *c, int n_seq, const uint8_t *const *seqs, const uint64_t *lens)
{
    if (!c || n_seq < 0 || (n_seq > 0 && (!seqs || !lens))) return fail(c, SNK_E_ARG, "bad arguments");
    const int rc = upload_impl(c, n_seq, seqs, lens);
    if (rc != SNK_OK) {                      // a failed upload leaves no sequences resident and no device memory behind
        (void)hipStreamSynchronize(c->stream);
        free_sequences(c);
    }
    return rc;
}

static int upload_impl(snk_ctx *c, int n_seq, const uint8_t *const *seqs, const uint64_t *lens)
{
    const auto up_t0 = std::chrono::steady_clock::now();
    auto up_last = up_t0;
    auto up_lap = [&]() {                                   // host wall time since the previous lap (every stage ends with a stream sync)
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - up_last).count();
        up_last = now;
        return ms;
    };
    for (double &v : c->up_ms) v = 0.0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->jobs_in_flight) { HIPCHK(c, hipEventSynchronize(c->jobs_busy)); c->jobs_in_flight = false; }
    free_sequences(c);
    if (n_seq == 0) return SNK_OK;

    const size_t n = (size_t)n_seq;
    std::vector<size_t> boff(n), poff(n);
    size_t btot = SNK_PAD, ptot = SNK_ARENA_SLACK;  // the packed arena starts with a zero region (empty suffix + slack)
    for (size_t g = 0; g < n; ++g) {
        if (lens[g] >= 0x7E000000ull) return fail(c, SNK_E_TOOBIG, "sequence %zu too long (%llu B)", g, (unsigned long long)lens[g]);
        if (lens[g] && !seqs[g]) return fail(c, SNK_E_ARG, "sequence %zu is NULL", g);
        boff[g] = btot; btot += ((size_t)lens[g] + 63) / 64 * 64 + SNK_PAD;
    }
    c->len.resize(n);
    c->min_len = 0xFFFFFFFFu; c->max_len = 0u;
    for (size_t g = 0; g < n; ++g) {
        c->len[g] = (uint32_t)lens[g];
        c->min_len = std::min(c->min_len, c->len[g]); c->max_len = std::max(c->max_len, c->len[g]);
    }
    if (btot >= c->arena_limit)
        return fail(c, SNK_E_TOOBIG, "ASCII arena of %zu bytes exceeds the %llu-byte offset range of one upload "
                    "(callers split the matrix into blocks of sequences: snacc_amd.cli.blocked_sizes)", btot,
                    (unsigned long long)c->arena_limit);

    {   // The set's letter case, from a sample (up to 8 KiB from the middle of every sequence): the 2-bit kernel serves
        // ONE case per upload -- its slots are liblz4's hashes of the bytes -- and the other case's letters are exceptions
        // like any other byte (a soft-masked set goes by its majority).
        uint64_t up = 0, lo = 0;
        for (size_t g = 0; g < n; ++g) {
            const uint64_t m = std::min<uint64_t>(lens[g], 8192), s0 = (lens[g] - m) / 2;
            for (uint64_t i = s0; i < s0 + m; ++i) {
                const uint8_t ch = seqs[g][i];
                if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') up++;
                else if (ch == 'a' || ch == 'c' || ch == 'g' || ch == 't') lo++;
            }
        }
        const bool lower = lo > up;
        if (lower != c->lower) {
            c->lower = lower;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const int rc = upload_luts(c);
            if (rc != SNK_OK) return rc;
        }
    }
    const uint32_t lcase = c->lower ? 0x20u : 0u;                 // ORed into 'A' 'C' 'G' 'T': the set's four letters

    HIPCHK(c, hipMalloc((void **)&c->d_bytes, btot));
    HIPCHK(c, stage_upload(c->device, c->d_bytes, btot, n, seqs, lens, boff));       // every byte of the arena, padding included

    c->up_ms[0] = up_lap();

    // ---- classify: exception granules -----------------------------------------------------------
    // A sequence goes to the 2-bit kernel when it is pure upper-case ACGT, or when at most exc_limit of its
    // 16-base granules per 2^20 bases hold another byte (N runs, IUPAC codes): those few places are served by the
    // byte-accurate general path of that kernel (snk_fast.hip.h, "Exceptions").
    std::vector<uint32_t> foff(n, 0), fwords(n, 0);
    size_t ftot = 0;
    for (size_t g = 0; g < n; ++g) { fwords[g] = (uint32_t)((((size_t)lens[g] + 15) / 16 + 31) / 32 + 2); foff[g] = (uint32_t)ftot; ftot += fwords[g]; }
    DevTemp<uint32_t> t_raw, t_cnt;
    HIPCHK(c, hipMalloc((void **)&t_raw.p, std::max<size_t>(ftot, 1) * 4));
    HIPCHK(c, hipMalloc((void **)&t_cnt.p, n * sizeof(uint32_t)));
    uint32_t *const d_raw = t_raw.p, *const d_cnt = t_cnt.p;
    HIPCHK(c, hipMemsetAsync(d_raw, 0, std::max<size_t>(ftot, 1) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, n * sizeof(uint32_t), c->stream));
    // one descriptor per sequence for the segmented ingest kernels (snk_ingest.hip.h): ONE launch per pass over all sequences
    std::vector<SnkSeqDesc> desc(n);
    for (size_t g = 0; g < n; ++g) { desc[g].boff = (uint32_t)boff[g]; desc[g].len = (uint32_t)lens[g]; desc[g].foff = foff[g]; desc[g].poff = 0u; }
    DevTemp<SnkSeqDesc> t_desc;
    HIPCHK(c, hipMalloc((void **)&t_desc.p, n * sizeof(SnkSeqDesc)));
    HIPCHK(c, hipMemcpy(t_desc.p, desc.data(), n * sizeof(SnkSeqDesc), hipMemcpyHostToDevice));
    DevTemp<uint32_t> t_ids;                                // id lists of the passes that cover a subset: [0, n) sequences with
    HIPCHK(c, hipMalloc((void **)&t_ids.p, 3 * n * sizeof(uint32_t)));      // exceptions, [n, 2n) packed, [2n, 3n) hash set / slot stream
    // grid.x for a pass whose threads handle `unit` bytes each, over sequences of at most max_len bytes
    auto seg_grid_x = [&](uint64_t unit) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>((((uint64_t)c->max_len + unit - 1) / unit + 255) / 256, 1), 4096); };
    // launch `k` over the sequences ids[0..cnt) (ids == nullptr: all n), 65 535 rows of the grid at a time
#define SNK_SEG_LAUNCH(kern, unit, ids_ptr, cnt, ...)                                                        \
    for (size_t y0_ = 0; y0_ < (size_t)(cnt); y0_ += 65535u) {                                                \
        const uint32_t ny_ = (uint32_t)std::min<size_t>(65535u, (size_t)(cnt) - y0_);                          \
        hipLaunchKernelGGL(kern, dim3(seg_grid_x(unit), ny_), dim3(256), 0, c->stream, c->d_bytes, t_desc.p, \
                           (const uint32_t *)(ids_ptr), (uint32_t)y0_, __VA_ARGS__);                           \
    }
    SNK_SEG_LAUNCH(snk_excraw_seg_kernel, 16, nullptr, n, d_raw, d_cnt, lcase);
    HIPCHK(c, hipGetLastError());
    std::vector<uint32_t> ecount(n);
    HIPCHK(c, hipMemcpyAsync(ecount.data(), d_cnt, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    t_cnt.release();

    // ---- which sequences go to the 2-bit kernel ------------------------------------------------------
    // Candidates by flagged granules; then the exact runs of exception bytes ("sites") from the caller's bytes of
    // the flagged granules.  Measured against pure ACGT (tools/gpu_exc.py, 1 Mbp genomes): 100 / 300 / 1000 / 3000
    // scattered IUPAC codes per Mbp 75 / 58 / 37 / 20 %; 1 / 5 / 20 % lower case in runs of ~500 bases (soft-masked) 60 /
    // 23 / 15 %.  The byte kernels such sequences would go to run at 29 % (compact table: ACGT + N only), 14 % (<= 2048
    // distinct hashes) or 6 % (full table: several IUPAC codes, or soft-masked) of the 2-bit rate; at 40 / 60 % lower case
    // (a third / 45 % of the bases after the stretches' overlap) the 2-bit kernel still runs at 11 / 9 %.  So a sequence
    // stays on it up to 60 % of its 16-base granules flagged and 4 + 6144 sites per 2^20 bases; the case with fewer
    // letters is the flagged one (see the sample above).
    c->is_packed.assign(n, 0); c->has_exc.assign(n, 0); c->any_exc = false; c->any_other = false;
    std::vector<uint32_t> eoff(n, 0xFFFFFFFFu), runs, roff(n, 0);
    {
        // Round 4: with the other-case mode on a table in LDS the 2-bit kernel is faster than the byte kernels at EVERY density of
        // soft-masking measured (1024 x 1 Mbp, 30 / 40 / 60 % lower case: 107 k / 93 k / 79 k against 62 k pair-compr./s), so the share
        // of flagged granules no longer decides -- only the number of exception SITES does (every site costs general probes:
        // 4 + 1.25 * exc_limit runs per 2^20 bases, 2 564 per Mbp at the default; measured crossover for scattered IUPAC codes 2 400).
        std::vector<uint32_t> raw_all;                      // the raw granule flags of every sequence: ONE copy, when a candidate has any
        bool want_raw = false;
        for (size_t g = 0; g < n; ++g) {
            c->is_packed[g] = lens[g] > 0 && (ecount[g] == 0 || c->exc_limit > 0);
            want_raw |= c->is_packed[g] && ecount[g] != 0;
        }
        if (want_raw) {
            raw_all.resize(ftot);
            HIPCHK(c, hipMemcpy(raw_all.data(), d_raw, ftot * 4, hipMemcpyDeviceToHost));
        }
        // the exact runs of exception bytes of every candidate, from the caller's bytes of its flagged granules: host threads, one
        // sequence at a time each (a set of 60 % soft-masked genomes has 600 MB to look at)
        std::vector<std::vector<uint32_t>> seq_runs(n);
        std::vector<uint8_t> seq_other(n, 0), seq_ok(n, 0);
        {
            std::atomic<size_t> next(0);
            auto work = [&]() {
                for (;;) {
                    const size_t g = next.fetch_add(1);
                    if (g >= n) return;
                    if (!c->is_packed[g] || ecount[g] == 0) continue;
                    const uint32_t *raw = raw_all.data() + foff[g];
                    std::vector<uint32_t> &rr = seq_runs[g];
                    const uint64_t max_sites = 4u + (uint64_t)lens[g] * 160u / 1048576u * (uint64_t)c->exc_limit / 128u;
                    bool other = false, ok = true;
                    for (size_t w = 0; w < fwords[g] && ok; ++w) {
                        uint32_t bits = raw[w];
                        while (bits) {
                            const size_t gr = w * 32 + (size_t)__builtin_ctz(bits);
                            bits &= bits - 1;
                            for (size_t i = gr * 16; i < gr * 16 + 16 && i < lens[g]; ++i) {
                                const uint8_t ch = seqs[g][i];
                                if (ch == ('A' | lcase) || ch == ('C' | lcase) || ch == ('G' | lcase) || ch == ('T' | lcase)) continue;
                                { const uint8_t u = (uint8_t)(ch & ~0x20u); other |= (u == 'A' || u == 'C' || u == 'G' || u == 'T'); }
                                if (!rr.empty() && rr.back() == (uint32_t)i) rr.back() = (uint32_t)i + 1;       // extends the open run
                                else { rr.push_back((uint32_t)i); rr.push_back((uint32_t)i + 1); }
                            }
                        }
                        if (rr.size() / 2 > max_sites) ok = false;                 // too many: the byte kernel serves this one
                    }
                    seq_ok[g] = ok; seq_other[g] = other;
                }
            };
            const size_t nt = want_raw ? std::min<size_t>(16, std::max<size_t>(1, std::min<size_t>(n, std::thread::hardware_concurrency()))) : 1;
            std::vector<std::thread> ts;
            for (size_t t = 1; t < nt; ++t) ts.emplace_back(work);
            work();
            for (auto &th : ts) th.join();
        }
        for (size_t g = 0; g < n; ++g) {
            if (!c->is_packed[g] || ecount[g] == 0) continue;
            if (!seq_ok[g]) { c->is_packed[g] = 0; continue; }
            roff[g] = (uint32_t)(runs.size() / 2);
            runs.insert(runs.end(), seq_runs[g].begin(), seq_runs[g].end());
            runs.push_back(0xFFFFFFFFu); runs.push_back(0xFFFFFFFFu);
            c->has_exc[g] = 1; c->any_exc = true; eoff[g] = foff[g];
            c->any_other |= seq_other[g] != 0;
        }
    }
    c->up_ms[1] = up_lap();
    std::vector<uint32_t> ids_packed, ids_exc;              // the sequences of the pack pass / of the passes over sequences with exceptions
    for (size_t g = 0; g < n; ++g)
        if (c->is_packed[g]) {
            poff[g] = ptot; ptot += (((size_t)lens[g] + 3) / 4 + 63) / 64 * 64 + SNK_PAD; c->n_packed++;
            desc[g].poff = (uint32_t)poff[g];
            ids_packed.push_back((uint32_t)g);
            if (c->has_exc[g]) ids_exc.push_back((uint32_t)g);
        }
    ptot += SNK_ARENA_SLACK;
    if (ptot >= c->arena_limit)
        return fail(c, SNK_E_TOOBIG, "2-bit arena of %zu bytes exceeds the offset range of one upload", ptot);
    HIPCHK(c, hipMemcpy(t_desc.p, desc.data(), n * sizeof(SnkSeqDesc), hipMemcpyHostToDevice));      // (now with the packed offsets; the classify pass has ended)
    HIPCHK(c, hipMalloc((void **)&c->d_exc_off, n * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpy(c->d_exc_off, eoff.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (c->any_exc) {
        HIPCHK(c, hipMalloc((void **)&c->d_exc_flags, ftot * 4));
        HIPCHK(c, hipMemsetAsync(c->d_exc_flags, 0, ftot * 4, c->stream));
        HIPCHK(c, hipMemcpy(t_ids.p, ids_exc.data(), ids_exc.size() * 4, hipMemcpyHostToDevice));
        SNK_SEG_LAUNCH(snk_excdilate_seg_kernel, 16 * 32, t_ids.p, ids_exc.size(), (const uint32_t *)d_raw, c->d_exc_flags);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMalloc((void **)&c->d_exc_runs, runs.size() * 4));
        HIPCHK(c, hipMemcpy(c->d_exc_runs, runs.data(), runs.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void **)&c->d_exc_roff, n * 4));
        HIPCHK(c, hipMemcpy(c->d_exc_roff, roff.data(), n * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMalloc((void **)&c->d_packed, ptot));
    HIPCHK(c, hipMemsetAsync(c->d_packed, 0, ptot, c->stream));
    if (c->any_exc) {                 // the class arena: where the bytes of the 2-bit sequences are not the set's four letters
        HIPCHK(c, hipMalloc((void **)&c->d_pmask, ptot));
        HIPCHK(c, hipMemsetAsync(c->d_pmask, 0, ptot, c->stream));
        SNK_SEG_LAUNCH(snk_packmask_seg_kernel, 16, t_ids.p, ids_exc.size(), c->d_pmask, lcase);
        HIPCHK(c, hipGetLastError());
    }
    if (!ids_packed.empty()) {
        const bool all = ids_packed.size() == n;             // every sequence packed: no list
        if (!all) HIPCHK(c, hipMemcpy(t_ids.p + n, ids_packed.data(), ids_packed.size() * 4, hipMemcpyHostToDevice));
        SNK_SEG_LAUNCH(snk_pack_seg_kernel, 16, all ? nullptr : t_ids.p + n, ids_packed.size(), c->d_packed);
        HIPCHK(c, hipGetLastError());
    }

    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->up_ms[2] = up_lap();

    // ---- resident hash set: can the byte kernel use the compact table? --------------------------
    c->compact_ok = false; c->n_hashes = 0; c->compact_cap = 0;
    if (c->bytes_compact_opt != 0 && !c->bytes_legacy) {
        HIPCHK(c, hipMemsetAsync(c->d_hashset, 0, 128 * sizeof(uint32_t), c->stream));
        bool any_packed = false, any_bytes = false;
        std::vector<uint32_t> ids_h;
        for (size_t g = 0; g < n; ++g) {
            if (c->is_packed[g] && !c->force_generic) {
                any_packed = true;                                   // contributes the 894 ACGT hashes
                if (!c->has_exc[g]) continue;                        // ... and, with exceptions, the hashes of those places
            }
            if (lens[g] < 5) continue;
            any_bytes = true;
            ids_h.push_back((uint32_t)g);
        }
        if (!ids_h.empty()) {
            HIPCHK(c, hipMemcpy(t_ids.p + 2 * n, ids_h.data(), ids_h.size() * 4, hipMemcpyHostToDevice));
            SNK_SEG_LAUNCH(snk_hashset_seg_kernel, 1, t_ids.p + 2 * n, ids_h.size(), c->d_hashset);
        }
        HIPCHK(c, hipGetLastError());
        if (any_bytes || c->force_generic) {
            uint32_t set[128];
            HIPCHK(c, hipMemcpyAsync(set, c->d_hashset, sizeof set, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (any_packed) {
                std::vector<uint16_t> slot; std::vector<uint32_t> hash;
                build_luts(slot, hash, c->lower);
                for (uint32_t k = 0; k < 1024; ++k) set[hash[k] >> 5] |= 1u << (hash[k] & 31u);
            }
            std::vector<uint16_t> h2c(4096, 0xFFFF);
            int cnt = 0;
            for (uint32_t hsh = 0; hsh < 4096; ++hsh)
                if (set[hsh >> 5] >> (hsh & 31u) & 1u) { if (cnt < 2048) h2c[hsh] = (uint16_t)cnt; cnt++; }
            c->n_hashes = cnt;
            if (cnt <= 2048) {
                HIPCHK(c, hipMemcpy(c->d_lut_h2c, h2c.data(), 8192, hipMemcpyHostToDevice));
                c->compact_ok = true;
                c->compact_cap = cnt <= 1024 ? 1024 : 2048;
            }
        }
    }

    // ---- the same for the one-shot hash (sequences that can take part in a pair of <= 64 KiB) ---
    c->oneshot_compact_ok = false; c->n_hashes4 = 0;
    if (c->bytes_compact_opt != 0 && !c->bytes_legacy) {
        uint32_t *d_set4 = c->d_hashset + 128;
        HIPCHK(c, hipMemsetAsync(d_set4, 0, 256 * sizeof(uint32_t), c->stream));
        bool any = false;
        std::vector<uint32_t> ids_h;
        for (size_t g = 0; g < n; ++g) {
            if (lens[g] > SNK_BLOCK || lens[g] < 4) continue;
            any = true;
            ids_h.push_back((uint32_t)g);
        }
        if (any) {
            HIPCHK(c, hipStreamSynchronize(c->stream));              // (the list's third part may still be read by the pass above)
            HIPCHK(c, hipMemcpy(t_ids.p + 2 * n, ids_h.data(), ids_h.size() * 4, hipMemcpyHostToDevice));
            for (size_t y0 = 0; y0 < ids_h.size(); y0 += 65535u) {   // (sequences of <= 64 KiB: 256 blocks of 256 cover the longest)
                const uint32_t ny = (uint32_t)std::min<size_t>(65535u, ids_h.size() - y0);
                hipLaunchKernelGGL(snk_hashset4_seg_kernel, dim3(64, ny), dim3(256), 0, c->stream, c->d_bytes, t_desc.p,
                                   (const uint32_t *)(t_ids.p + 2 * n), (uint32_t)y0, d_set4);
            }
        }
        HIPCHK(c, hipGetLastError());
        if (any) {
            uint32_t set[256];
            HIPCHK(c, hipMemcpyAsync(set, d_set4, sizeof set, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::vector<uint16_t> h2c(8192, 0xFFFF);
            int cnt = 0;
            for (uint32_t hsh = 0; hsh < 8192; ++hsh)
                if (set[hsh >> 5] >> (hsh & 31u) & 1u) { if (cnt < 1024) h2c[hsh] = (uint16_t)cnt; cnt++; }
            c->n_hashes4 = cnt;
            if (cnt <= 1024) {
                HIPCHK(c, hipMemcpy(c->d_lut_h2c4, h2c.data(), 16384, hipMemcpyHostToDevice));
                c->oneshot_compact_ok = true;
            }
        }
    }

    // ---- slot stream of the byte kernels (linked mode) ------------------------------------------
    // needed when some pair of > 64 KiB cannot run on the 2-bit kernel: a sequence that is not packed, or force_generic
    {
        bool need = c->force_generic;
        for (size_t g = 0; g < n && !need; ++g) need = !c->is_packed[g];
        if (need && !c->bytes_legacy && c->max_len > 0 && (uint64_t)c->max_len * 2u > SNK_BLOCK) {
            HIPCHK(c, hipMalloc((void **)&c->d_slots, btot * sizeof(uint16_t)));
            HIPCHK(c, hipMemsetAsync(c->d_slots, 0, btot * sizeof(uint16_t), c->stream));
            SNK_SEG_LAUNCH(snk_slotstream_seg_kernel, 1, nullptr, n,
                           c->compact_ok ? (const uint16_t *)c->d_lut_h2c : (const uint16_t *)nullptr, c->d_slots);
            HIPCHK(c, hipGetLastError());
        }
    }
#undef SNK_SEG_LAUNCH
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->up_ms[3] = up_lap();

    // ---- per-sequence tables ------------------------------------------------------------------
    std::vector<const uint8_t *> bp(n); std::vector<uint32_t> pp(n), bo(n);
    std::vector<uint32_t> spos(n);
    for (size_t g = 0; g < n; ++g) {
        bp[g] = c->d_bytes + boff[g]; bo[g] = (uint32_t)boff[g];
        pp[g] = c->is_packed[g] ? (uint32_t)poff[g] : 0u;
        spos[g] = lens[g] > SNK_BLOCK ? (uint32_t)(lens[g] / SNK_BLOCK * SNK_BLOCK) : 0u;
    }
    HIPCHK(c, hipMalloc((void **)&c->d_bytes_ptr, n * sizeof(void *)));
    HIPCHK(c, hipMalloc((void **)&c->d_packed_off, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&c->d_bytes_off, n * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&c->d_len, n * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_snap_pos, n * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_snap_out, n * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_single, n * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_snap_fast, n * SNK_FSLOTS * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_snap_gen, n * 4096 * 4));
    HIPCHK(c, hipMemcpy(c->d_bytes_ptr, bp.data(), n * sizeof(void *), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_packed_off, pp.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_bytes_off, bo.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->boff = bo;
    HIPCHK(c, hipMemcpy(c->d_len, c->len.data(), n * 4, hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> order(n);
        for (size_t g = 0; g < n; ++g) order[g] = (uint32_t)g;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c->len[a] > c->len[b]; });
        HIPCHK(c, hipMalloc((void **)&c->d_yorder, n * 4));
        HIPCHK(c, hipMemcpy(c->d_yorder, order.data(), n * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemcpy(c->d_snap_pos, spos.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemsetAsync(c->d_snap_out, 0, n * 4, c->stream));
    c->n = n_seq;

    c->single_have.assign(n, 0);
    c->singles_done = true;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->up_ms[4] = up_lap();
    if (!c->defer_singles) {
        const int rc = ensure_singles(c, 0, n_seq);        // phase A for every sequence, as part of the upload
        if (rc) return rc;
    }
    c->up_ms[5] = c->defer_singles ? 0.0 : up_lap();
    c->up_ms[6] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - up_t0).count();
    return SNK_OK;
}

int snk_singles_rows(snk_ctx *c, int r0, int r1, uint32_t *sizes)
{
    if (!c) return SNK_E_ARG;
    if (r0 < 0 || r1 < r0 || r1 > c->n || (!sizes && r1 > r0)) return fail(c, SNK_E_ARG, "bad row range [%d,%d)", r0, r1);
    if (!c->singles_done && c->n) return fail(c, SNK_E_STATE, "snk_upload has not completed");
    if (r1 == r0) return SNK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = ensure_singles(c, r0, r1);              // (option defer_singles: phase A of these rows runs now)
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(sizes, c->d_single + r0, (size_t)(r1 - r0) * 4, hipMemcpyDeviceToHost));
    return SNK_OK;
}

int snk_singles(snk_ctx *c, uint32_t *sizes)
{
    if (!c || (!sizes && c->n)) return fail(c, SNK_E_ARG, "bad arguments");
    return snk_singles_rows(c, 0, c->n, sizes);
}

int snk_upload_times(const snk_ctx *c, double *ms, int cap)
{
    if (!c || cap < 0 || (cap && !ms)) return SNK_E_ARG;
    for (int k = 0; k < cap && k < 7; ++k) ms[k] = c->up_ms[k];
    return 7;
}

int snk_pairs_device(snk_ctx *c, int r0, int r1, void *d_sizes, void *hip_stream)
{
    if (!c) return SNK_E_ARG;
    if (!c->singles_done) return fail(c, SNK_E_STATE, "snk_upload has not completed");
    if (r0 < 0 || r1 < r0 || r1 > c->n || (!d_sizes && r1 > r0)) return fail(c, SNK_E_ARG, "bad row range [%d,%d)", r0, r1);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const size_t N = (size_t)c->n, np = (size_t)(r1 - r0) * N;
    if (!np) return SNK_OK;
    if ((uint64_t)np > 0xFFF00000ull)          // job numbers and output indices are 32-bit on the device (and the job counter overshoots a little)
        return fail(c, SNK_E_TOOBIG, "%zu ordered pairs in one launch (rows [%d,%d) x %zu): at most 0xFFF00000; tile the rows", np, r0, r1, N);
    size_t nf = 0, nb = 0, ng = 0;
    int rc = ensure_singles(c, r0, r1);        // the prefix snapshots of these rows (a no-op unless phase A was deferred)
    if (rc) return rc;
    // (a set in which some sequences carry exceptions and others do not goes through the list: the clean pairs run on the pure kernel)
    bool some_clean = false;
    if (c->split_clean && c->any_exc) {
        size_t rows_cl = 0, cols_cl = 0;
        for (size_t g = 0; g < (size_t)c->n; ++g) {
            const bool cl = c->is_packed[g] && !c->has_exc[g];
            cols_cl += cl; rows_cl += cl && (int)g >= r0 && (int)g < r1;
        }
        some_clean = rows_cl && (c->split_clean == 2 || rows_cl * cols_cl >= split_threshold(c));
    }
    if (!c->force_generic && !some_clean && c->n_packed == c->n && (uint64_t)c->min_len * 2u > SNK_BLOCK && (uint64_t)c->max_len * 2u < 0x7E000000ull) {
        nf = np; c->dense_tile = true; c->n_fast_clean = 0;             // every pair fits the 2-bit kernel: no job list at all
    } else {
        // order: suffix j outer, prefix i inner => the chains of a wave share seq_j
        rc = build_jobs(c, np, [&](size_t t, int &i, int &j, uint32_t &o) {
            j = (int)(t / (size_t)(r1 - r0)); i = r0 + (int)(t % (size_t)(r1 - r0));
            o = (uint32_t)((size_t)(i - r0) * N + (size_t)j);
        }, nf, nb, ng);
        if (rc) return rc;
    }
    SnkTileDesc tile;
    if (c->dense_tile) {      // every pair goes to the 2-bit kernel: the kernel derives the pair from the job number
        tile.r0 = (uint32_t)r0; tile.rows = (uint32_t)(r1 - r0); tile.n = (uint32_t)N;
    }
    // ragged lengths (a job costs ~ tail of x + len y): hand batches out dynamically
    tile.ragged = (uint64_t)c->max_len * 10u > (uint64_t)c->min_len * 11u;
    rc = ensure_scratch(c, tile.rows ? 0 : np, 0);
    if (rc) return rc;
    return run_pairs(c, st, nf, nb, ng, (uint32_t *)d_sizes, &tile);
}

int snk_sync(snk_ctx *c, void *hip_stream)
{
    if (!c) return SNK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : c->stream));
    int rc = check_status(c);
    if (rc) return rc;
    return c->dfl ? snk_internal_dfl_check(c) : SNK_OK;
}

int snk_pairs(snk_ctx *c, int r0, int r1, uint32_t *sizes)
{
    if (!c) return SNK_E_ARG;
    if (r0 < 0 || r1 < r0 || r1 > c->n || (!sizes && r1 > r0)) return fail(c, SNK_E_ARG, "bad row range [%d,%d)", r0, r1);
    if (r1 == r0 || c->n == 0) return SNK_OK;
    // Row tiles of bounded size: the job list (16 B per pair) and the device output stay small
    // however large N is; a tile still holds whole rows so one workgroup's chains share a suffix.
    const size_t N = (size_t)c->n;
    const size_t max_pairs = (size_t)4 << 20;
    // equal tiles (no small last one: a tile that does not fill the card costs as much as one that does)
    const size_t max_rows = std::max<size_t>(1, max_pairs / N);
    const size_t n_tiles = ((size_t)(r1 - r0) + max_rows - 1) / max_rows;
    const int tile_rows = (int)(((size_t)(r1 - r0) + n_tiles - 1) / n_tiles);
    double ms_total = 0.0;
    for (int t0 = r0; t0 < r1; t0 += tile_rows) {
        const int t1 = std::min(r1, t0 + tile_rows);
        const size_t np = (size_t)(t1 - t0) * N;
        int rc = ensure_scratch(c, 0, np);
        if (rc) return rc;
        rc = snk_pairs_device(c, t0, t1, c->d_out, nullptr);
        if (rc) return rc;
        rc = snk_sync(c, nullptr);
        if (rc) return rc;
        HIPCHK(c, hipMemcpy(sizes + (size_t)(t0 - r0) * N, c->d_out, np * 4, hipMemcpyDeviceToHost));
        const double ms = snk_last_pairs_ms(c);
        if (ms > 0) ms_total += ms;
    }
    c->ms_accum = ms_total;
    return SNK_OK;
}

int snk_pairs_list(snk_ctx *c, int n_pairs, const int32_t *ij, uint32_t *sizes)
{
    if (!c || n_pairs < 0 || (n_pairs && (!ij || !sizes))) return fail(c, SNK_E_ARG, "bad arguments");
    if (!c->singles_done) return fail(c, SNK_E_STATE, "snk_upload has not completed");
    if (!n_pairs) return SNK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    size_t nf = 0, nb = 0, ng = 0;
    {   // the prefix snapshots the list needs (a no-op unless phase A was deferred)
        std::vector<uint32_t> rows;
        std::vector<uint8_t> seen((size_t)c->n, 0);
        for (int t = 0; t < n_pairs; ++t) {
            const int i = ij[2 * t];
            if (i < 0 || i >= c->n) return fail(c, SNK_E_ARG, "pair index out of range");
            if (!seen[(size_t)i] && !c->single_have[(size_t)i]) { seen[(size_t)i] = 1; rows.push_back((uint32_t)i); }
        }
        const int rc0 = run_singles(c, rows);
        if (rc0) return rc0;
    }
    // sort by suffix for locality, keep the caller's output order
    std::vector<uint32_t> order((size_t)n_pairs);
    for (uint32_t t = 0; t < (uint32_t)n_pairs; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ij[2 * a + 1] < ij[2 * b + 1]; });
    int rc = build_jobs(c, (size_t)n_pairs, [&](size_t t, int &i, int &j, uint32_t &o) {
        o = order[t]; i = ij[2 * o]; j = ij[2 * o + 1];
    }, nf, nb, ng);
    if (rc) return rc;
    rc = ensure_scratch(c, (size_t)n_pairs, (size_t)n_pairs);
    if (rc) return rc;
    rc = run_pairs(c, c->stream, nf, nb, ng, c->d_out);
    if (rc) return rc;
    rc = snk_sync(c, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(sizes, c->d_out, (size_t)n_pairs * 4, hipMemcpyDeviceToHost));
    return SNK_OK;
}

#ifdef SNK_TRACE
/* diagnostic build only (not part of the shipped ABI) */
int snk_debug_trace(unsigned int from, unsigned int *out /* [1 + 4*4096] */, int read)
{
    if (!read) {
        unsigned int zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(snk_trace_n), &zero, 4) != hipSuccess) return -1;
        return hipMemcpyToSymbol(HIP_SYMBOL(snk_trace_from), &from, 4) == hipSuccess ? 0 : -1;
    }
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(snk_trace_n), 4) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out + 1, HIP_SYMBOL(snk_trace_buf), 4 * 4096 * 4) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SNK_STATS
/* diagnostic build only (not part of the shipped ABI): read and clear the event counters */
int snk_debug_stats(unsigned long long *out16 /* [64] */)
{
    unsigned long long zero[64] = { 0 };
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(snk_stats), sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(snk_stats), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

#ifdef SNK_STAMP
/* diagnostic build only (not part of the shipped ABI) */
int snk_debug_read_stamps(unsigned long long *out8)
{
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(snk_stamp_buf), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

int snk_frames_list(snk_ctx *c, int n_items, const int32_t *ij, const uint64_t *offsets, uint8_t *out)
{
    if (!c || n_items < 0 || (n_items && (!ij || !offsets || !out))) return fail(c, SNK_E_ARG, "bad arguments");
    if (!c->singles_done) return fail(c, SNK_E_STATE, "snk_upload has not completed");
    if (!n_items) return SNK_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<SnkEmitJob> jobs((size_t)n_items);
    for (int t = 0; t < n_items; ++t) {
        const int i = ij[2 * t], j = ij[2 * t + 1];
        if (i < 0 || i >= c->n || j < -1 || j >= c->n) return fail(c, SNK_E_ARG, "item %d: index out of range", t);
        const uint64_t n = (uint64_t)c->len[i] + (j >= 0 ? c->len[j] : 0u);
        if (n >= 0x7E000000ull) return fail(c, SNK_E_TOOBIG, "item %d too long", t);
        if (offsets[t + 1] < offsets[t]) return fail(c, SNK_E_ARG, "offsets must be non-decreasing");
        jobs[(size_t)t].xi = i; jobs[(size_t)t].yi = j; jobs[(size_t)t].off = offsets[t];
    }
    const uint64_t total = offsets[n_items];
    SnkEmitJob *d_jobs = nullptr; uint8_t *d_frames = nullptr; uint32_t *d_sizes = nullptr;
    auto cleanup = [&]() { dfree(d_jobs); dfree(d_frames); dfree(d_sizes); };
#define EMCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail(c, SNK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
    EMCHK(hipMalloc((void **)&d_jobs, jobs.size() * sizeof(SnkEmitJob)));
    EMCHK(hipMalloc((void **)&d_frames, total + 64));
    EMCHK(hipMalloc((void **)&d_sizes, (size_t)n_items * 4));
    EMCHK(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(SnkEmitJob), hipMemcpyHostToDevice, c->stream));
    const uint32_t chains = 8;
    const size_t lds = (size_t)chains * 16384;
    EMCHK(hipFuncSetAttribute((const void *)snk_emit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(snk_emit_kernel, dim3((uint32_t)((n_items + chains - 1) / chains)), dim3(64), lds, c->stream,
                       make_tables(c), d_jobs, (uint32_t)n_items, chains, d_frames, d_sizes, c->d_status);
    EMCHK(hipGetLastError());
    EMCHK(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> sizes((size_t)n_items);
    EMCHK(hipMemcpy(sizes.data(), d_sizes, sizes.size() * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < n_items; ++t)
        if ((uint64_t)sizes[(size_t)t] != offsets[t + 1] - offsets[t]) {
            cleanup();
            return fail(c, SNK_E_ARG, "item %d: frame is %u bytes but the offsets reserve %llu", t, sizes[(size_t)t],
                        (unsigned long long)(offsets[t + 1] - offsets[t]));
        }
    EMCHK(hipMemcpy(out, d_frames, total, hipMemcpyDeviceToHost));
#undef EMCHK
    cleanup();
    return check_status(c);
}

int snk_pairs_ms_log(snk_ctx *c, double *ms, int cap)
{
    if (!c || cap < 0 || (cap && !ms)) return SNK_E_ARG;
    c->ev_logging = true;                        // from now on every launch keeps its own event pair until the next read
    const size_t n = c->ev_used;
    int k = 0;
    for (size_t t = 0; t < n; ++t) {
        float v = 0.f;
        if (hipEventElapsedTime(&v, c->ev_log[t].first, c->ev_log[t].second) != hipSuccess) {
            c->ev_used = 0; c->ev_valid = false;
            return fail(c, SNK_E_STATE, "a logged launch has not completed (call snk_sync first)");
        }
        if (k < cap) ms[k] = (double)v;
        k++;
    }
    c->ev_used = 0; c->ev_valid = false;
    return k;
}

double snk_last_pairs_ms(snk_ctx *c)
{
    if (!c || !c->ev_valid) return -1.0;
    if (c->ms_accum >= 0.0) { const double v = c->ms_accum; return v; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

} // extern "C"
