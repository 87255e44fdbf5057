// snk_deflate.hip -- gzip / zlib compressed SIZES on gfx950.
//
// Replaces, for the batched path, the codec calls of
//   ref:snacc/pairwise_ncd.py:73-74   gzip.compress(sequence)   (zlib deflate level 9)
//   ref:snacc/pairwise_ncd.py:77-78   zlib.compress(sequence)   (zlib deflate level 6)
// with bit-exact sizes of the raw deflate stream zlib 1.2.11 writes (deflate_slow, memLevel 8,
// 32 KiB window); the 18 / 6 wrapper bytes and sys.getsizeof's 33 are added in Python.
//
// How the work is organised:
//  * zlib inserts EVERY position into its hash chains at these levels, so the chain of a position
//    is a pure function of the data: "earlier positions with the same 15-bit hash of 3 bytes, most
//    recent first".  Per resident sequence a stable radix sort by hash gives that list with random
//    access (occ / inv2 / bucket starts, plus the 8 bytes at every listed position); a wave then tests up
//    to 256 chain candidates per step, in chain order, honouring zlib's chain budget, distance limits,
//    nice-length early exit and "first longest wins".  For level 9 a second list keyed by six bytes
//    narrows the candidates to the ones that can win.
//  * one wavefront = one parse job; everything deflate_slow decides (lazy evaluation, TOO_FAR,
//    block cut every 16383 symbols, stored / static / dynamic choice with zlib's exact
//    build_tree / gen_bitlen / scan_tree) is done by the wave; the trees are built on VGPR lane
//    arrays with scalar control (small alphabets) or by lane 0 in LDS (more than 64 used symbols).
//  * the same argument as below parallelises the per-sequence pass: 32 KiB segments, stitched where
//    two parsers provably agree.
//  * exact work elision: a match can reach back 32 506 bytes only, so the symbol stream of x+y
//    equals x's own stream until just before the seam and y's own stream from the first point,
//    at least 32 507 bytes after the seam, where both parsers stand right behind a match ending
//    at the same position.  A pair job therefore restarts from x's stream ~600 bytes before the
//    seam, parses ~33 K positions, finds that point, and from there only re-cuts y's stored
//    symbols into blocks (the block phase differs per pair) and prices them.
#include "snk_internal.h"
#include "snacc_hip.h"

#include <hipcub/hipcub.hpp>

#include "snk_deflate.hip.h"

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct DflState {
    int device = 0;
    int n = 0;
    int level = 0;                       // level the symbol streams were built for (0 = none)
    bool indexed = false, kindexed = false;
    std::vector<DflSeq> seq;
    DflSeq *d_seq = nullptr;
    uint32_t *d_occ = nullptr, *d_bstart = nullptr;
    uint64_t *d_occ8 = nullptr, *d_inv2 = nullptr;
    DflKRec *d_krec = nullptr;
    uint32_t *d_kbstart = nullptr;
    uint64_t *d_kinv2 = nullptr;
    uint32_t *d_sym = nullptr, *d_pos = nullptr, *d_rhist = nullptr, *d_status = nullptr, *d_chk = nullptr;
    uint64_t *d_cumbits = nullptr;
    DflJob *d_jobs = nullptr; size_t jobs_cap = 0;
    uint32_t *d_out = nullptr; size_t out_cap = 0;
    std::vector<uint32_t> single;        // raw stream bytes of every sequence at `level`
    uint32_t *d_seg_sym = nullptr, *d_seg_pos = nullptr, *d_seg_cnt = nullptr;   // scratch of the per-sequence pass
    int n_serial = 0;                     // sequences of the last per-sequence pass that needed the serial parse
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = -1.0;               // device time of the parse kernels of the last pairs call
    bool pending = false;                // an asynchronous launch whose event pair has not been read yet
};

template <typename P> void dfree(P *&p) { if (p) { (void)hipFree((void *)p); p = nullptr; } }

// Temporaries of one build step: freed on every way out of the scope (the early returns of DCHK included).
struct DevTemps {
    std::vector<void **> slots;
    template <typename P> void own(P *&p) { slots.push_back((void **)&p); }
    ~DevTemps() { for (void **q : slots) if (*q) { (void)hipFree(*q); *q = nullptr; } }
};

void dfl_destroy(void *v)
{
    DflState *s = (DflState *)v;
    if (!s) return;
    (void)hipSetDevice(s->device);
    dfree(s->d_seq); dfree(s->d_occ); dfree(s->d_occ8); dfree(s->d_inv2); dfree(s->d_bstart);
    dfree(s->d_krec); dfree(s->d_kbstart); dfree(s->d_kinv2); dfree(s->d_sym); dfree(s->d_pos);
    dfree(s->d_rhist); dfree(s->d_status); dfree(s->d_cumbits); dfree(s->d_jobs); dfree(s->d_out);
    dfree(s->d_seg_sym); dfree(s->d_seg_pos); dfree(s->d_seg_cnt); dfree(s->d_chk);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

#define DCHK(c, call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char m_[384];                                                                    \
            snprintf(m_, sizeof m_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return snk_internal_fail((c), SNK_E_HIP, m_);                                    \
        }                                                                                    \
    } while (0)

bool level_config(int level, DflTables &T)
{
    if (level == 9) { T.good = 32; T.lazy = 258; T.nice = 258; T.chain = 4096; return true; }
    if (level == 6) { T.good = 8; T.lazy = 16; T.nice = 128; T.chain = 128; return true; }
    return false;
}

DflTables make_tables(const DflState *s, const SnkSeqView &v, int level)
{
    DflTables T{};
    T.bytes = v.d_bytes; T.seq = s->d_seq; T.occ = s->d_occ; T.occ8 = s->d_occ8; T.inv2 = s->d_inv2; T.bstart = s->d_bstart;
    T.sym = s->d_sym; T.pos = s->d_pos; T.cumbits = s->d_cumbits; T.rhist = s->d_rhist; T.status = s->d_status;
    T.seg_sym = s->d_seg_sym; T.seg_pos = s->d_seg_pos; T.seg_cnt = s->d_seg_cnt;
    T.k_rec = s->d_krec; T.k_bstart = s->d_kbstart; T.k_inv2 = s->d_kinv2;
    T.chk = s->d_chk;
    T.norestart = v.dfl_norestart ? 1u : 0u;
    T.use_k = (v.dfl_kmer && level == 9 && s->kindexed) ? 1u : 0u;   // pays off only for the 4096-member budget
    level_config(level, T);
    return T;
}

int dfl_get(snk_ctx *c, SnkSeqView &v, DflState *&s)
{
    int rc = snk_internal_view(c, &v);
    if (rc != SNK_OK) return rc;
    if (v.n <= 0) return snk_internal_fail(c, SNK_E_STATE, "no sequences resident (call snk_upload first)");
    void (**free_fn)(void *) = nullptr;
    void **slot = snk_internal_dfl_slot(c, &free_fn);
    if (!*slot) {
        DflState *ns = new (std::nothrow) DflState();
        if (!ns) return snk_internal_fail(c, SNK_E_HIP, "out of host memory");
        ns->device = v.device; ns->n = v.n;
        *slot = ns; *free_fn = dfl_destroy;
    }
    s = (DflState *)*slot;
    return SNK_OK;
}

// Tear the add-on's state down (after a failed build): the next call starts from nothing.
void dfl_drop(snk_ctx *c)
{
    void (**free_fn)(void *) = nullptr;
    void **slot = snk_internal_dfl_slot(c, &free_fn);
    if (*slot) { dfl_destroy(*slot); *slot = nullptr; }
}

// Device-side checks of the deflate kernels (DFL_ST_* bits): read, clear, and turn into SNK_E_KERNEL.
int dfl_check_status(snk_ctx *c, DflState *s)
{
    if (!s || !s->d_status) return SNK_OK;
    uint32_t st = 0;
    DCHK(c, hipMemcpy(&st, s->d_status, 4, hipMemcpyDeviceToHost));
    if (st) {
        (void)hipMemset(s->d_status, 0, 4);
        char m[128];
        snprintf(m, sizeof m, "deflate kernel: device-side check failed (status 0x%x: symbol stream beyond its capacity)", st);
        return snk_internal_fail(c, SNK_E_KERNEL, m);
    }
    return SNK_OK;
}

int dfl_ensure_scratch(snk_ctx *c, DflState *s, size_t njobs)
{
    if (njobs > s->jobs_cap) {
        dfree(s->d_jobs);
        DCHK(c, hipMalloc((void **)&s->d_jobs, njobs * sizeof(DflJob)));
        s->jobs_cap = njobs;
    }
    if (njobs > s->out_cap) {
        dfree(s->d_out);
        DCHK(c, hipMalloc((void **)&s->d_out, njobs * sizeof(uint32_t)));
        s->out_cap = njobs;
    }
    return SNK_OK;
}

// Launch nj jobs that are already in s->d_jobs, on `stream`, results to d_out (device).  Brackets the kernel with
// the state's event pair; does not synchronise.
int dfl_launch_dev(snk_ctx *c, DflState *s, const SnkSeqView &v, int level, uint32_t nj, bool seg, hipStream_t stream, uint32_t *d_out)
{
    const DflTables T = make_tables(s, v, level);
    if (!s->ev0) { DCHK(c, hipEventCreate(&s->ev0)); DCHK(c, hipEventCreate(&s->ev1)); }
    DCHK(c, hipEventRecord(s->ev0, stream));
    const dim3 grid((nj + DFL_WAVES - 1u) / DFL_WAVES), block(64 * DFL_WAVES);
    if (T.use_k && seg)       hipLaunchKernelGGL((dfl_parse_kernel_k<true>), grid, block, L_GROUP, stream, T, s->d_jobs, nj, d_out);
    else if (T.use_k)         hipLaunchKernelGGL((dfl_parse_kernel_k<false>), grid, block, L_GROUP, stream, T, s->d_jobs, nj, d_out);
    else if (seg)             hipLaunchKernelGGL((dfl_parse_kernel<true>), grid, block, L_GROUP, stream, T, s->d_jobs, nj, d_out);
    else                      hipLaunchKernelGGL((dfl_parse_kernel<false>), grid, block, L_GROUP, stream, T, s->d_jobs, nj, d_out);
    DCHK(c, hipGetLastError());
    DCHK(c, hipEventRecord(s->ev1, stream));
    return SNK_OK;
}

int dfl_add_elapsed(DflState *s)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->last_ms = (s->last_ms < 0 ? 0.0 : s->last_ms) + (double)ms;
    return SNK_OK;
}

int dfl_launch(snk_ctx *c, DflState *s, const SnkSeqView &v, int level, const std::vector<DflJob> &jobs, uint32_t *host_out)
{
    if (jobs.empty()) return SNK_OK;
    int rc = dfl_ensure_scratch(c, s, jobs.size());
    if (rc != SNK_OK) return rc;
    DCHK(c, hipMemcpyAsync(s->d_jobs, jobs.data(), jobs.size() * sizeof(DflJob), hipMemcpyHostToDevice, v.stream));
    rc = dfl_launch_dev(c, s, v, level, (uint32_t)jobs.size(), jobs[0].mode == 2u, v.stream, s->d_out);   // a launch is all segment jobs or none
    if (rc != SNK_OK) return rc;
    if (host_out) {
        DCHK(c, hipMemcpyAsync(host_out, s->d_out, jobs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, v.stream));
    }
    DCHK(c, hipStreamSynchronize(v.stream));
    rc = dfl_check_status(c, s);
    if (rc != SNK_OK) return rc;
    return dfl_add_elapsed(s);
}

// Rows [r0, r1) x all columns: the job list is written by a kernel, nothing but the sizes crosses the bus.
int dfl_launch_rows(snk_ctx *c, DflState *s, const SnkSeqView &v, int level, size_t r0, size_t r1, hipStream_t stream, uint32_t *d_out)
{
    const size_t nj = (r1 - r0) * (size_t)v.n;
    if (nj == 0) return SNK_OK;
    if (nj > 0xFFFFFFF0ull) return snk_internal_fail(c, SNK_E_ARG, "too many pairs in one call");
    if (nj > s->jobs_cap) {
        DCHK(c, hipStreamSynchronize(stream));
        dfree(s->d_jobs);
        DCHK(c, hipMalloc((void **)&s->d_jobs, nj * sizeof(DflJob)));
        s->jobs_cap = nj;
    }
    hipLaunchKernelGGL(dfl_rowjobs_kernel, dim3((uint32_t)std::min<size_t>((nj + 255) / 256, 8192)), dim3(256), 0, stream,
                       s->d_jobs, (uint32_t)r0, (uint32_t)(r1 - r0), (uint32_t)v.n);
    DCHK(c, hipGetLastError());
    return dfl_launch_dev(c, s, v, level, (uint32_t)nj, false, stream, d_out);
}

int dfl_build_index(snk_ctx *c, DflState *s, const SnkSeqView &v)
{
    const size_t n = (size_t)v.n;
    s->seq.assign(n, DflSeq{});
    uint64_t itot = 0, stot = 0, ctot = 0;
    uint32_t maxlen = 0;
    for (size_t g = 0; g < n; ++g) {
        DflSeq &q = s->seq[g];
        q.boff = v.boff[g]; q.len = v.len[g];
        q.ioff = itot; itot += (uint64_t)q.len + 1u;
        q.soff = stot; stot += (uint64_t)q.len + 1u;
        q.coff = ctot; ctot += (uint64_t)q.len / DFL_BLOCK_SYMS + 2u;
        maxlen = std::max(maxlen, q.len);
    }
    DCHK(c, hipMalloc((void **)&s->d_seq, n * sizeof(DflSeq)));
    DCHK(c, hipMalloc((void **)&s->d_occ, itot * 4));
    DCHK(c, hipMalloc((void **)&s->d_occ8, itot * 8));
    DCHK(c, hipMalloc((void **)&s->d_inv2, itot * 8));
    DCHK(c, hipMalloc((void **)&s->d_bstart, n * (DFL_NHASH + 1u) * 4));
    DCHK(c, hipMalloc((void **)&s->d_sym, stot * 4));
    DCHK(c, hipMalloc((void **)&s->d_pos, stot * 4));
    DCHK(c, hipMalloc((void **)&s->d_cumbits, ctot * 8));
    DCHK(c, hipMalloc((void **)&s->d_rhist, n * DFL_HIST * 4));
    DCHK(c, hipMalloc((void **)&s->d_status, 4));
    DCHK(c, hipMemsetAsync(s->d_status, 0, 4, v.stream));
    DCHK(c, hipMemsetAsync(s->d_inv2, 0, itot * 8, v.stream));

    uint16_t *d_key = nullptr, *d_skey = nullptr; uint32_t *d_val = nullptr; void *d_tmp = nullptr; size_t tmp_bytes = 0;
    DevTemps temps; temps.own(d_key); temps.own(d_skey); temps.own(d_val); temps.own(d_tmp);
    const size_t cap = (size_t)maxlen + 1u;
    DCHK(c, hipMalloc((void **)&d_key, cap * 2));
    DCHK(c, hipMalloc((void **)&d_skey, cap * 2));
    DCHK(c, hipMalloc((void **)&d_val, cap * 4));
    {
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, d_skey, d_val, s->d_occ, (int)cap, 0, 16, v.stream);
        if (e != hipSuccess) return snk_internal_fail(c, SNK_E_HIP, "radix sort sizing failed");
    }
    DCHK(c, hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    for (size_t g = 0; g < n; ++g) {
        const DflSeq &q = s->seq[g];
        const uint32_t m = q.len >= 3u ? q.len - 2u : 0u;
        uint32_t *bst = s->d_bstart + g * (DFL_NHASH + 1u);
        if (m) {
            const uint32_t grid = std::min<uint32_t>((m + 255u) / 256u, 4096u);
            hipLaunchKernelGGL(dfl_hash_kernel, dim3(grid), dim3(256), 0, v.stream, v.d_bytes + q.boff, m, d_key, d_val);
            hipError_t e = hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_skey, d_val, s->d_occ + q.ioff, (int)m, 0, 15, v.stream);
            if (e != hipSuccess) return snk_internal_fail(c, SNK_E_HIP, "radix sort failed");
        }
        hipLaunchKernelGGL(dfl_bstart_kernel, dim3((DFL_NHASH + 1u + 255u) / 256u), dim3(256), 0, v.stream, d_skey, m, bst);
        if (m) {
            const uint32_t grid = std::min<uint32_t>((m + 255u) / 256u, 4096u);
            hipLaunchKernelGGL(dfl_inv_kernel, dim3(grid), dim3(256), 0, v.stream, s->d_occ + q.ioff, d_skey, bst,
                               v.d_bytes + q.boff, m, s->d_inv2 + q.ioff, s->d_occ8 + q.ioff);
        }
    }
    DCHK(c, hipGetLastError());
    DCHK(c, hipStreamSynchronize(v.stream));
    s->indexed = true;
    return SNK_OK;
}

// The six-byte index (level 9 only; built on first use).
int dfl_build_kindex(snk_ctx *c, DflState *s, const SnkSeqView &v)
{
    const size_t n = (size_t)v.n;
    uint64_t itot = 0;
    uint32_t maxlen = 0;
    for (const DflSeq &q : s->seq) { itot += (uint64_t)q.len + 1u; maxlen = std::max(maxlen, q.len); }
    DCHK(c, hipMalloc((void **)&s->d_krec, itot * sizeof(DflKRec)));
    DCHK(c, hipMalloc((void **)&s->d_kinv2, itot * 8));
    DCHK(c, hipMalloc((void **)&s->d_kbstart, n * 65537u * 4));
    uint16_t *d_key = nullptr, *d_skey = nullptr; uint32_t *d_val = nullptr, *d_kocc = nullptr; void *d_tmp = nullptr; size_t tmp_bytes = 0;
    DevTemps temps; temps.own(d_key); temps.own(d_skey); temps.own(d_val); temps.own(d_kocc); temps.own(d_tmp);
    const size_t cap = (size_t)maxlen + 1u;
    DCHK(c, hipMalloc((void **)&d_key, cap * 2));
    DCHK(c, hipMalloc((void **)&d_skey, cap * 2));
    DCHK(c, hipMalloc((void **)&d_val, cap * 4));
    DCHK(c, hipMalloc((void **)&d_kocc, cap * 4));            // one sequence's positions by bucket (the records take them over)
    {
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, d_skey, d_val, d_kocc, (int)cap, 0, 16, v.stream);
        if (e != hipSuccess) return snk_internal_fail(c, SNK_E_HIP, "radix sort sizing failed");
    }
    DCHK(c, hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    for (size_t g = 0; g < n; ++g) {
        const DflSeq &q = s->seq[g];
        // the six-byte index (positions <= len - 6), same construction
        const uint32_t m6 = q.len >= 6u ? q.len - 5u : 0u;
        uint32_t *kbst = s->d_kbstart + g * 65537u;
        if (m6) {
            const uint32_t grid = std::min<uint32_t>((m6 + 255u) / 256u, 4096u);
            hipLaunchKernelGGL(dfl_khash_kernel, dim3(grid), dim3(256), 0, v.stream, v.d_bytes + q.boff, m6, d_key, d_val);
            hipError_t e = hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_skey, d_val, d_kocc, (int)m6, 0, 16, v.stream);
            if (e != hipSuccess) return snk_internal_fail(c, SNK_E_HIP, "radix sort failed");
        }
        hipLaunchKernelGGL(dfl_kbstart_kernel, dim3((65537u + 255u) / 256u), dim3(256), 0, v.stream, d_skey, m6, kbst);
        if (m6) {
            const uint32_t grid = std::min<uint32_t>((m6 + 255u) / 256u, 4096u);
            hipLaunchKernelGGL(dfl_kinv_kernel, dim3(grid), dim3(256), 0, v.stream, d_kocc, d_skey, kbst,
                               v.d_bytes + q.boff, s->d_inv2 + q.ioff, m6, s->d_kinv2 + q.ioff, s->d_krec + q.ioff);
        }
    }
    DCHK(c, hipGetLastError());
    DCHK(c, hipStreamSynchronize(v.stream));
    s->kindexed = true;
    return SNK_OK;
}

int dfl_prepare(snk_ctx *c, int level, SnkSeqView &v, DflState *&s)
{
    DflTables cfg{};
    if (!level_config(level, cfg)) return snk_internal_fail(c, SNK_E_ARG, "deflate level must be 9 (gzip) or 6 (zlib)");
    int rc = dfl_get(c, v, s);
    if (rc != SNK_OK) return rc;
    DCHK(c, hipSetDevice(v.device));
    // a build that fails half way (out of memory is the realistic case: ~52 B per base at level 9) is torn
    // down completely, so that a retry starts from nothing instead of allocating over the old pointers
    if (!s->indexed) { rc = dfl_build_index(c, s, v); if (rc != SNK_OK) { dfl_drop(c); s = nullptr; return rc; } }
    if (level == 9 && v.dfl_kmer && !s->kindexed) {
        rc = dfl_build_kindex(c, s, v);
        if (rc != SNK_OK) { dfree(s->d_krec); dfree(s->d_kinv2); dfree(s->d_kbstart); return rc; }
    }
    if (s->level == level) return SNK_OK;
    // stand-alone stream of every sequence at this level
    dfree(s->d_chk);
    for (auto &q : s->seq) { q.nsym = 0; q.unsafe = 0; q.total_bits = 0; q.rk = q.rpos = q.rkb = q.rbpos = 0; }
    // (every copy and memset below is ordered on v.stream: the kernels run there, and it does not synchronise
    // with the null stream)
    DCHK(c, hipMemcpyAsync(s->d_seq, s->seq.data(), s->seq.size() * sizeof(DflSeq), hipMemcpyHostToDevice, v.stream));
    DCHK(c, hipStreamSynchronize(v.stream));
    s->single.assign((size_t)v.n, 0u);
    std::vector<int> serial;                 // sequences parsed by one wave from start to end
    if (v.dfl_serial) {
        for (int g = 0; g < v.n; ++g) serial.push_back(g);
    } else {
        // ---- in parallel segments, stitched where two parsers provably agree (dfl_stitch_kernel) ----
        std::vector<DflSeg> segs;
        std::vector<int> seg_seq;
        uint64_t aux = 0;
        for (int g = 0; g < v.n; ++g) {
            const uint32_t len = v.len[g];
            const uint32_t k = std::max<uint32_t>(1u, (len + DFL_SEG - 1u) / DFL_SEG);
            for (uint32_t i = 0; i < k; ++i) {
                DflSeg sg{};
                sg.p0 = i * DFL_SEG; sg.p1 = std::min<uint64_t>((uint64_t)(i + 1u) * DFL_SEG, len);
                sg.first = i == 0u; sg.aux = aux; sg.cnt = 0;
                aux += (uint64_t)(sg.p1 - sg.p0) + DFL_SEG_SLACK + DFL_SEG_ROOM;
                segs.push_back(sg); seg_seq.push_back(g);
            }
        }
        const size_t ns = segs.size();
        dfree(s->d_seg_sym); dfree(s->d_seg_pos); dfree(s->d_seg_cnt);
        DCHK(c, hipMalloc((void **)&s->d_seg_sym, aux * 4));
        DCHK(c, hipMalloc((void **)&s->d_seg_pos, aux * 4));
        DCHK(c, hipMalloc((void **)&s->d_seg_cnt, ns * 4));
        DflSeg *d_segs = nullptr; uint32_t *d_from = nullptr, *d_ends = nullptr, *d_fail = nullptr, *d_num = nullptr; uint64_t *d_dst = nullptr;
        DevTemps temps; temps.own(d_segs); temps.own(d_from); temps.own(d_ends); temps.own(d_fail); temps.own(d_num); temps.own(d_dst);
        DCHK(c, hipMalloc((void **)&d_segs, ns * sizeof(DflSeg)));
        DCHK(c, hipMalloc((void **)&d_from, ns * 4)); DCHK(c, hipMalloc((void **)&d_ends, ns * 4));
        DCHK(c, hipMalloc((void **)&d_fail, ns * 4)); DCHK(c, hipMalloc((void **)&d_num, ns * 4));
        DCHK(c, hipMalloc((void **)&d_dst, ns * 8));
        DCHK(c, hipMemcpyAsync(d_segs, segs.data(), ns * sizeof(DflSeg), hipMemcpyHostToDevice, v.stream));
        DCHK(c, hipMemsetAsync(d_fail, 0, ns * 4, v.stream));
        DCHK(c, hipStreamSynchronize(v.stream));
        std::vector<DflJob> jobs(ns);
        for (size_t t = 0; t < ns; ++t) jobs[t] = DflJob{seg_seq[t], -1, 2u, (uint32_t)t, segs[t].p0, segs[t].p1, segs[t].aux};
        rc = dfl_launch(c, s, v, level, jobs, nullptr);
        if (rc != SNK_OK) return rc;
        hipLaunchKernelGGL(dfl_stitch_kernel, dim3((uint32_t)((ns + 63) / 64)), dim3(64), 0, v.stream, d_segs, s->d_seg_cnt,
                           (uint32_t)ns, s->d_seg_sym, s->d_seg_pos, d_from, d_ends, d_fail);
        DCHK(c, hipGetLastError());
        std::vector<uint32_t> cnt(ns), from(ns), ends(ns), failv(ns), num(ns, 0u);
        std::vector<uint64_t> dst(ns, 0ull);
        DCHK(c, hipMemcpyAsync(cnt.data(), s->d_seg_cnt, ns * 4, hipMemcpyDeviceToHost, v.stream));
        DCHK(c, hipMemcpyAsync(from.data(), d_from, ns * 4, hipMemcpyDeviceToHost, v.stream));
        DCHK(c, hipMemcpyAsync(ends.data(), d_ends, ns * 4, hipMemcpyDeviceToHost, v.stream));
        DCHK(c, hipMemcpyAsync(failv.data(), d_fail, ns * 4, hipMemcpyDeviceToHost, v.stream));
        DCHK(c, hipStreamSynchronize(v.stream));
        std::vector<DflJob> price;
        for (size_t t = 0; t < ns;) {
            const int g = seg_seq[t];
            size_t e = t;
            while (e < ns && seg_seq[e] == g) ++e;
            bool ok = true;
            uint64_t total = 0;
            for (size_t u = t; u < e && ok; ++u) {
                const uint32_t hi = u + 1 < e ? ends[u] : cnt[u];      // the last segment keeps everything to its end
                if (failv[u] || hi < from[u]) { ok = false; break; }
                num[u] = hi - from[u]; dst[u] = s->seq[(size_t)g].soff + total; total += num[u];
            }
            if (ok && total <= (uint64_t)v.len[g] + 1u) {
                s->seq[(size_t)g].nsym = (uint32_t)total;
                price.push_back(DflJob{g, -1, 3u, (uint32_t)price.size(), 0u, 0u, 0ull});
            } else {
                for (size_t u = t; u < e; ++u) num[u] = 0;
                serial.push_back(g);                                   // e.g. periodic data whose parsers never meet
            }
            t = e;
        }
        DCHK(c, hipMemcpyAsync(d_num, num.data(), ns * 4, hipMemcpyHostToDevice, v.stream));
        DCHK(c, hipMemcpyAsync(d_dst, dst.data(), ns * 8, hipMemcpyHostToDevice, v.stream));
        hipLaunchKernelGGL(dfl_compact_kernel, dim3((uint32_t)ns), dim3(256), 0, v.stream, d_segs, d_from, d_num, d_dst,
                           s->d_seg_sym, s->d_seg_pos, s->d_sym, s->d_pos);
        DCHK(c, hipGetLastError());
        DCHK(c, hipMemcpyAsync(s->d_seq, s->seq.data(), s->seq.size() * sizeof(DflSeq), hipMemcpyHostToDevice, v.stream));
        DCHK(c, hipStreamSynchronize(v.stream));
        std::vector<uint32_t> sizes(price.size());
        rc = dfl_launch(c, s, v, level, price, sizes.data());        // out_idx = sequence index: sized below
        if (rc != SNK_OK) return rc;
        dfree(d_segs); dfree(d_from); dfree(d_ends); dfree(d_fail); dfree(d_num); dfree(d_dst);
        dfree(s->d_seg_sym); dfree(s->d_seg_pos); dfree(s->d_seg_cnt);
        s->n_serial = (int)serial.size();
    }
    if (!serial.empty()) {
        std::vector<DflJob> jobs(serial.size());
        for (size_t t = 0; t < serial.size(); ++t) jobs[t] = DflJob{serial[t], -1, 1u, (uint32_t)t, 0u, 0u, 0ull};
        rc = dfl_launch(c, s, v, level, jobs, nullptr);
        if (rc != SNK_OK) return rc;
    }
    const DflTables T = make_tables(s, v, level);
    hipLaunchKernelGGL(dfl_restart_kernel, dim3((uint32_t)v.n), dim3(64), 0, v.stream, T, (uint32_t)v.n);
    DCHK(c, hipGetLastError());
    DCHK(c, hipStreamSynchronize(v.stream));
    DCHK(c, hipMemcpyAsync(s->seq.data(), s->d_seq, s->seq.size() * sizeof(DflSeq), hipMemcpyDeviceToHost, v.stream));
    DCHK(c, hipStreamSynchronize(v.stream));
    for (int g = 0; g < v.n; ++g) s->single[(size_t)g] = (uint32_t)(s->seq[(size_t)g].total_bits >> 3);
    // cumulative code counts at every 128th symbol of every stored stream: a pair job prices y's blocks from two
    // checkpoints and at most 254 symbols instead of streaming all of y's symbols
    {
        uint64_t ktot = 0;
        for (auto &q : s->seq) { q.koff = ktot; ktot += ((uint64_t)q.nsym / DFL_CHK + 1u) * DFL_HIST; }
        dfree(s->d_chk);
        DCHK(c, hipMalloc((void **)&s->d_chk, ktot * 4));
        DCHK(c, hipMemcpyAsync(s->d_seq, s->seq.data(), s->seq.size() * sizeof(DflSeq), hipMemcpyHostToDevice, v.stream));
        DflTables Tc = make_tables(s, v, level);
        hipLaunchKernelGGL(dfl_chk_kernel, dim3((uint32_t)v.n), dim3(64), 0, v.stream, Tc, s->d_chk, (uint32_t)v.n);
        DCHK(c, hipGetLastError());
        DCHK(c, hipStreamSynchronize(v.stream));
    }
    s->level = level;
    return SNK_OK;
}

}  // namespace

extern "C" {

#ifdef DFL_STAMP
/* diagnostic build only: the stored symbol stream of sequence g (cap entries) and its restart record */
int snk_debug_dfl_stream(snk_ctx *c, int g, uint32_t *sym, uint32_t *pos, uint32_t cap, uint32_t *info /* [8] */)
{
    void **slot = snk_internal_dfl_slot(c, nullptr);
    DflState *s = (DflState *)*slot;
    if (!s || g < 0 || g >= s->n) return -1;
    DflSeq q;
    if (hipMemcpy(&q, s->d_seq + g, sizeof q, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const uint32_t k = q.nsym < cap ? q.nsym : cap;
    if (hipMemcpy(sym, s->d_sym + q.soff, (size_t)k * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(pos, s->d_pos + q.soff, (size_t)k * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    info[0] = q.nsym; info[1] = q.unsafe; info[2] = q.rk; info[3] = q.rpos; info[4] = q.rkb; info[5] = q.rbpos;
    info[6] = (uint32_t)(q.total_bits >> 3); info[7] = (uint32_t)s->n_serial;
    return 0;
}
int snk_debug_dfl_stamps(unsigned long long *out /* [64*8] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dfl_stamp_buf), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -1;
}
#endif

int snk_deflate_prepare(snk_ctx *c, int level)
{
    if (!c) return SNK_E_ARG;
    SnkSeqView v; DflState *s = nullptr;
    return dfl_prepare(c, level, v, s);
}

int snk_deflate_singles(snk_ctx *c, int level, uint32_t *sizes)
{
    if (!c || !sizes) return c ? snk_internal_fail(c, SNK_E_ARG, "sizes is NULL") : SNK_E_ARG;
    SnkSeqView v; DflState *s = nullptr;
    int rc = dfl_prepare(c, level, v, s);
    if (rc != SNK_OK) return rc;
    std::copy(s->single.begin(), s->single.end(), sizes);
    return SNK_OK;
}

int snk_deflate_pairs(snk_ctx *c, int level, int row_begin, int row_end, uint32_t *sizes)
{
    if (!c || !sizes) return c ? snk_internal_fail(c, SNK_E_ARG, "sizes is NULL") : SNK_E_ARG;
    SnkSeqView v; DflState *s = nullptr;
    int rc = dfl_prepare(c, level, v, s);
    if (rc != SNK_OK) return rc;
    if (row_begin < 0 || row_end > v.n || row_begin > row_end) return snk_internal_fail(c, SNK_E_ARG, "row range out of bounds");
    const size_t n = (size_t)v.n;
    s->last_ms = -1.0;
    // in tiles of about a million pairs
    const size_t tile_rows = std::max<size_t>(1, (size_t)(1u << 20) / n);
    for (size_t r0 = (size_t)row_begin; r0 < (size_t)row_end; r0 += tile_rows) {
        const size_t r1 = std::min<size_t>((size_t)row_end, r0 + tile_rows);
        const size_t nj = (r1 - r0) * n;
        if (nj > s->out_cap) {
            dfree(s->d_out);
            DCHK(c, hipMalloc((void **)&s->d_out, nj * sizeof(uint32_t)));
            s->out_cap = nj;
        }
        rc = dfl_launch_rows(c, s, v, level, r0, r1, v.stream, s->d_out);
        if (rc != SNK_OK) return rc;
        DCHK(c, hipMemcpyAsync(sizes + (r0 - (size_t)row_begin) * n, s->d_out, nj * sizeof(uint32_t), hipMemcpyDeviceToHost, v.stream));
        DCHK(c, hipStreamSynchronize(v.stream));
        dfl_add_elapsed(s);
    }
    return dfl_check_status(c, s);
}

/* Same, asynchronous: rows [row_begin, row_end) on `hip_stream` (NULL = the context's stream), u32 raw stream
 * sizes to DEVICE memory d_sizes ((row_end - row_begin) * n_seq elements); no synchronisation (snk_sync). */
int snk_deflate_pairs_device(snk_ctx *c, int level, int row_begin, int row_end, void *d_sizes, void *hip_stream)
{
    if (!c || !d_sizes) return c ? snk_internal_fail(c, SNK_E_ARG, "d_sizes is NULL") : SNK_E_ARG;
    SnkSeqView v; DflState *s = nullptr;
    int rc = dfl_prepare(c, level, v, s);
    if (rc != SNK_OK) return rc;
    if (row_begin < 0 || row_end > v.n || row_begin > row_end) return snk_internal_fail(c, SNK_E_ARG, "row range out of bounds");
    s->last_ms = -1.0;
    s->pending = true;
    return dfl_launch_rows(c, s, v, level, (size_t)row_begin, (size_t)row_end, hip_stream ? (hipStream_t)hip_stream : v.stream, (uint32_t *)d_sizes);
}

int snk_deflate_pairs_list(snk_ctx *c, int level, int n_pairs, const int32_t *ij, uint32_t *sizes)
{
    if (!c || n_pairs < 0 || (n_pairs > 0 && (!ij || !sizes))) return c ? snk_internal_fail(c, SNK_E_ARG, "bad arguments") : SNK_E_ARG;
    SnkSeqView v; DflState *s = nullptr;
    int rc = dfl_prepare(c, level, v, s);
    if (rc != SNK_OK) return rc;
    s->last_ms = -1.0;
    std::vector<DflJob> jobs((size_t)n_pairs);
    for (int t = 0; t < n_pairs; ++t) {
        const int32_t i = ij[2 * t], j = ij[2 * t + 1];
        if (i < 0 || i >= v.n || j < -1 || j >= v.n) return snk_internal_fail(c, SNK_E_ARG, "pair index out of range");
        jobs[(size_t)t] = DflJob{i, j, 0u, (uint32_t)t, 0u, 0u, 0ull};
    }
    return dfl_launch(c, s, v, level, jobs, sizes);
}

/* Device time (ms, hipEvent pair around the parse kernels) of the last snk_deflate_pairs /
 * snk_deflate_pairs_list call; < 0 if unavailable. */
double snk_deflate_last_ms(snk_ctx *c)
{
    if (!c) return -1.0;
    void **slot = snk_internal_dfl_slot(c, nullptr);
    DflState *s = (DflState *)*slot;
    if (!s) return -1.0;
    if (s->pending && s->ev1) {                           // asynchronous launch: valid once its stream has been synchronised
        if (hipEventSynchronize(s->ev1) != hipSuccess) return -1.0;
        s->pending = false;
        dfl_add_elapsed(s);
    }
    return s->last_ms;
}

}  // extern "C"

// snk_sync (snacc_hip.hip) ends with this: the asynchronous deflate launches report through their own status word.
int snk_internal_dfl_check(snk_ctx *c)
{
    void (**free_fn)(void *) = nullptr;
    void **slot = snk_internal_dfl_slot(c, &free_fn);
    return *slot ? dfl_check_status(c, (DflState *)*slot) : SNK_OK;
}
