/*
 * oracle/lz4f_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (size only) of the compressor behind the reference's lz4 hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product path (snacc_amd/csrc) never links or calls it.
 *
 * What it restates
 * ----------------
 * ref:snacc/pairwise_ncd.py:80   compressed_seq = lz4framed.compress(sequence)
 * ref:snacc/pairwise_ncd.py:90   sys.getsizeof(compressed_seq)   (= len + 33, added in Python)
 *
 * `lz4framed` is the third-party wheel py-lz4framed (ref:requirements.txt:6,
 * unpinned; ref:README.md:74 shows 0.12.0).  It is absent from /root/reference
 * and from this image.  It wraps liblz4's LZ4F_compressFrame; this file restates
 * the PUBLISHED algorithm of liblz4 1.9.3 (LZ4 frame format + LZ4 "fast" block
 * compressor, level 0, acceleration 1) for LZ4F_compressFrame(prefs = NULL):
 *   - 64 KiB blocks, no checksums, no content-size field,
 *   - srcSize <= 64 KiB: one independent block, one-shot compressor
 *     (u16 table, 8192 entries, 13-bit hash of 4 bytes, no distance check),
 *   - srcSize  > 64 KiB: linked blocks, streaming compressor (u32 table of
 *     4096 absolute positions shared by all blocks, 12-bit hash of 5 bytes,
 *     64 KiB-1 distance limit),
 *   - each block compressed with limitedOutput (dstCapacity = blockLen-1); on
 *     overflow the block is stored raw AND the hash table keeps only the
 *     insertions made before the overflow was detected.
 *
 * Pinning: tests/test_oracle_liblz4.py differential-fuzzes this file against
 * the liblz4 1.9.3 binary of this image (dlopen, test-only) and against the
 * committed golden vectors in tests/golden/ (SURVEY.md 8c).  Parity against a
 * real py-lz4framed wheel is UNPINNED (unknown bundled liblz4 version; unknown
 * content-size flag => possible constant +8 B): see DESIGN.md.
 */
#define _POSIX_C_SOURCE 200809L      /* clock_gettime (timed CPU baseline) under -std=c11 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define SNK_BLOCK      65536u
#define MINMATCH       4u
#define MFLIMIT        12u
#define LASTLITERALS   5u
#define MINLENGTH      (MFLIMIT + 1u)
#define MAXDIST        65535u
#define SKIP_TRIGGER   6u

static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* 12-bit hash of the 5 bytes at p (little endian) -- linked/streaming mode */
static inline uint32_t hash5(const uint8_t *p)
{
    return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> 52);
}
/* 13-bit hash of the 4 bytes at p -- one-shot mode (input < 64 KiB + 11) */
static inline uint32_t hash4(const uint8_t *p)
{
    return (rd32(p) * 2654435761U) >> 19;
}

static inline uint32_t lit_ext(uint32_t lit) { return lit >= 15u ? (lit - 15u) / 255u + 1u : 0u; }

/* statistics, optional (tests use them to sanity-check kernels) */
typedef struct {
    uint64_t sequences, search_probes, chain_probes, too_far, back_ext, bailouts;
} snk_oracle_stats;

/*
 * One LZ4 block [start, start+len) of the stream `base`, compressed with
 * limitedOutput and maxOutputSize = len-1.
 *   linked != 0 : table32 holds absolute stream positions, hash5/12-bit,
 *                 distance check, matches may reach back to stream start.
 *   linked == 0 : table16 (start must be 0), hash4/13-bit, no distance check.
 * Returns the compressed size, or 0 when the block does not fit in len-1 bytes
 * (caller stores it raw).  The table is left exactly as liblz4 leaves it.
 */
static uint32_t block_size(const uint8_t *base, uint32_t start, uint32_t len,
                           int linked, uint32_t *table32, uint16_t *table16,
                           snk_oracle_stats *st)
{
    const uint32_t iend = start + len;
    const uint32_t olimit = len - 1u;          /* len >= 1 here */
    uint32_t op = 0;
    uint32_t ip = start, anchor = start;

#define HASH(pos)     (linked ? hash5(base + (pos)) : hash4(base + (pos)))
#define TGET(h)       (linked ? table32[h] : (uint32_t)table16[h])
#define TPUT(h, pos)  do { if (linked) table32[h] = (pos); else table16[h] = (uint16_t)(pos); } while (0)

    if (len < MINLENGTH) goto last_literals;

    {
        const uint32_t mfl1 = iend - MFLIMIT + 1u;   /* mflimitPlusOne */
        const uint32_t mlimit = iend - LASTLITERALS; /* matchlimit     */
        uint32_t fh, cand;

        TPUT(HASH(ip), ip);
        ip++;
        fh = HASH(ip);

        for (;;) {
            /* ---- search for a match ---- */
            {
                uint32_t fip = ip, step = 1, nb = 1u << SKIP_TRIGGER;
                for (;;) {
                    uint32_t h = fh;
                    uint32_t cur = fip;
                    cand = TGET(h);
                    ip = fip;
                    fip += step;
                    step = nb++ >> SKIP_TRIGGER;
                    if (fip > mfl1) goto last_literals;
                    fh = HASH(fip);
                    TPUT(h, cur);
                    if (st) st->search_probes++;
                    if (linked && cand + MAXDIST < cur) { if (st) st->too_far++; continue; }
                    if (rd32(base + cand) == rd32(base + ip)) break;
                }
            }
            /* ---- catch up (backward extension) ---- */
            while (ip > anchor && cand > 0 && base[ip - 1] == base[cand - 1]) {
                ip--; cand--;
                if (st) st->back_ext++;
            }
            /* ---- literals ---- */
            {
                uint32_t lit = ip - anchor;
                op++;                                          /* token */
                if (op + lit + (2u + 1u + LASTLITERALS) + lit / 255u > olimit) {
                    if (st) st->bailouts++;
                    return 0;
                }
                op += lit_ext(lit) + lit;
            }
            for (;;) {
                /* ---- _next_match: offset + match length ---- */
                uint32_t mc = 0;
                op += 2;
                {
                    uint32_t a = ip + MINMATCH, b = cand + MINMATCH;
                    while (a < mlimit && base[a] == base[b]) { a++; b++; }
                    mc = a - (ip + MINMATCH);
                }
                ip += mc + MINMATCH;
                if (op + (1u + LASTLITERALS) + (mc + 240u) / 255u > olimit) {
                    if (st) st->bailouts++;
                    return 0;
                }
                if (mc >= 15u) op += (mc - 15u) / 255u + 1u;
                if (st) st->sequences++;
                anchor = ip;
                if (ip >= mfl1) goto last_literals;

                TPUT(HASH(ip - 2), ip - 2);
                {
                    uint32_t h = HASH(ip);
                    cand = TGET(h);
                    TPUT(h, ip);
                    if (st) st->chain_probes++;
                    if ((!linked || cand + MAXDIST >= ip) && rd32(base + cand) == rd32(base + ip)) {
                        op++;                                  /* token, zero literals */
                        continue;
                    }
                }
                break;
            }
            ip++;
            fh = HASH(ip);
        }
    }

last_literals:
    {
        uint32_t run = iend - anchor;
        if (op + run + 1u + (run + 255u - 15u) / 255u > olimit) {
            if (st) st->bailouts++;
            return 0;
        }
        op += 1u + lit_ext(run) + run;
    }
    return op;
#undef HASH
#undef TGET
#undef TPUT
}

/* ---------------------------------------------------------------------------
 * Streaming state (linked mode) -- exposed so tests can check the exact-reuse
 * properties the GPU path relies on (prefix snapshots).
 * ------------------------------------------------------------------------- */
typedef struct {
    uint32_t table[4096];
    uint64_t pos;      /* bytes consumed  */
    uint64_t out;      /* frame bytes so far, header included */
} snk_oracle_stream;

void snk_oracle_stream_init(snk_oracle_stream *s)
{
    memset(s, 0, sizeof(*s));
    s->out = 7;  /* magic(4) FLG BD HC */
}

/* Compress blocks of `base[0..n)` (n > 65536: linked mode) from s->pos up to
 * `upto` (a block boundary or n).  Returns 0, or -1 on bad arguments. */
int snk_oracle_stream_run(snk_oracle_stream *s, const uint8_t *base, uint64_t n,
                          uint64_t upto, snk_oracle_stats *st)
{
    if (n >= 0x7E000000ull || upto > n || (s->pos % SNK_BLOCK) != 0) return -1;
    while (s->pos < upto) {
        uint32_t start = (uint32_t)s->pos;
        uint32_t len = (uint32_t)((n - s->pos) < SNK_BLOCK ? (n - s->pos) : SNK_BLOCK);
        uint32_t c = block_size(base, start, len, 1, s->table, NULL, st);
        s->out += 4u + (c ? c : len);
        s->pos += len;
    }
    return 0;
}

/* Whole-frame size of LZ4F_compressFrame(dst, cap, p, n, NULL). */
uint64_t snk_oracle_lz4f_size(const uint8_t *p, uint64_t n)
{
    if (n >= 0x7E000000ull) return 0;          /* liblz4 renormalises near 2 GiB: out of scope */
    if (n == 0) return 11;
    if (n <= SNK_BLOCK) {
        uint16_t *t16 = (uint16_t *)calloc(8192, sizeof(uint16_t));
        uint32_t c;
        if (!t16) return 0;
        c = block_size(p, 0, (uint32_t)n, 0, NULL, t16, NULL);
        free(t16);
        return 7u + 4u + (c ? c : (uint32_t)n) + 4u;
    } else {
        snk_oracle_stream *s = (snk_oracle_stream *)malloc(sizeof(*s));
        uint64_t r;
        if (!s) return 0;
        snk_oracle_stream_init(s);
        snk_oracle_stream_run(s, p, n, n, NULL);
        r = s->out + 4u;                         /* end mark */
        free(s);
        return r;
    }
}

/* Same, with workload statistics (linked mode only; n > 65536). */
uint64_t snk_oracle_lz4f_size_stats(const uint8_t *p, uint64_t n, snk_oracle_stats *st)
{
    snk_oracle_stream *s;
    uint64_t r;
    memset(st, 0, sizeof(*st));
    if (n <= SNK_BLOCK || n >= 0x7E000000ull) return snk_oracle_lz4f_size(p, n);
    s = (snk_oracle_stream *)malloc(sizeof(*s));
    if (!s) return 0;
    snk_oracle_stream_init(s);
    snk_oracle_stream_run(s, p, n, n, st);
    r = s->out + 4u;
    free(s);
    return r;
}

/* Frame size of the concatenation x+y without the caller materialising it twice
 * (ref:snacc/pairwise_ncd.py:29-30 concatenates the two extracted sequences). */
uint64_t snk_oracle_lz4f_size_pair(const uint8_t *x, uint64_t nx, const uint8_t *y, uint64_t ny)
{
    uint8_t *buf = (uint8_t *)malloc(nx + ny + 8);
    uint64_t r;
    if (!buf) return 0;
    memcpy(buf, x, nx);
    memcpy(buf + nx, y, ny);
    r = snk_oracle_lz4f_size(buf, nx + ny);
    free(buf);
    return r;
}

/* ---------------------------------------------------------------------------
 * Synthetic genome generator of SURVEY.md 8c (portable 64-bit LCG).
 * ------------------------------------------------------------------------- */
void snk_oracle_lcg_genome(uint64_t seed, uint64_t n, uint8_t *out)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    uint64_t s = seed, i;
    for (i = 0; i < n; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        out[i] = (uint8_t)acgt[(s >> 33) & 3];
    }
}

/* 2 % point-mutant of `src` (LCG seed over positions, substitute
 * when (s>>40)%50==0 with "ACGT"[(s>>33)&3]). */
void snk_oracle_lcg_mutant(const uint8_t *src, uint64_t seed, uint64_t n, uint8_t *out)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    uint64_t s = seed, i;
    for (i = 0; i < n; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        out[i] = ((s >> 40) % 50 == 0) ? (uint8_t)acgt[(s >> 33) & 3] : src[i];
    }
}

/* ---------------------------------------------------------------------------
 * Bounded multi-thread CPU baseline helper for bench.py (cpu_baseline leg):
 * frame sizes of all ordered pairs (i,j), i in [r0,r1), j in [0,n) of `n`
 * equally-addressed sequences.  Threads split the pair list; plain pthreads.
 * ------------------------------------------------------------------------- */
#include <pthread.h>
typedef struct {
    const uint8_t *const *seqs; const uint64_t *lens; int n, r0, r1;
    uint32_t *out; int tid, nthreads;
} pair_job;

static void *pair_worker(void *arg)
{
    pair_job *j = (pair_job *)arg;
    long total = (long)(j->r1 - j->r0) * j->n, k;
    for (k = j->tid; k < total; k += j->nthreads) {
        int a = j->r0 + (int)(k / j->n), b = (int)(k % j->n);
        j->out[k] = (uint32_t)snk_oracle_lz4f_size_pair(j->seqs[a], j->lens[a], j->seqs[b], j->lens[b]);
    }
    return NULL;
}

int snk_oracle_pairs_mt(const uint8_t *const *seqs, const uint64_t *lens, int n,
                        int r0, int r1, uint32_t *out, int nthreads)
{
    pthread_t *th;
    pair_job *jobs;
    int t;
    if (nthreads < 1) nthreads = 1;
    th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    jobs = (pair_job *)malloc(sizeof(pair_job) * nthreads);
    if (!th || !jobs) return -1;
    for (t = 0; t < nthreads; t++) {
        jobs[t] = (pair_job){ seqs, lens, n, r0, r1, out, t, nthreads };
        pthread_create(&th[t], NULL, pair_worker, &jobs[t]);
    }
    for (t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

/* Frame sizes of an arbitrary list of ordered pairs ij[2t], ij[2t+1] (tests: random samples of
 * the full-size matrix). */
typedef struct {
    const uint8_t *const *seqs; const uint64_t *lens; const int32_t *ij; long n_pairs;
    uint32_t *out; int tid, nthreads;
} list_job;

static void *list_worker(void *arg)
{
    list_job *j = (list_job *)arg;
    long k;
    for (k = j->tid; k < j->n_pairs; k += j->nthreads) {
        int a = j->ij[2 * k], b = j->ij[2 * k + 1];
        j->out[k] = (uint32_t)snk_oracle_lz4f_size_pair(j->seqs[a], j->lens[a], j->seqs[b], j->lens[b]);
    }
    return NULL;
}

int snk_oracle_pairs_list_mt(const uint8_t *const *seqs, const uint64_t *lens, const int32_t *ij,
                             long n_pairs, uint32_t *out, int nthreads)
{
    pthread_t *th;
    list_job *jobs;
    int t;
    if (nthreads < 1) nthreads = 1;
    th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    jobs = (list_job *)malloc(sizeof(list_job) * nthreads);
    if (!th || !jobs) return -1;
    for (t = 0; t < nthreads; t++) {
        jobs[t] = (list_job){ seqs, lens, ij, n_pairs, out, t, nthreads };
        pthread_create(&th[t], NULL, list_worker, &jobs[t]);
    }
    for (t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

/* ---------------------------------------------------------------------------
 * Timed CPU baseline for bench.py (cpu_baseline leg, kind "port"): every thread owns ONE
 * concatenation buffer and ONE stream state, allocated before the clock starts, and compresses
 * ordered pairs (a, b) of the given sequences until `seconds` have passed.  Reports the pairs
 * completed by all threads and the wall time of the slowest thread.
 * ------------------------------------------------------------------------- */
#include <time.h>
typedef struct {
    const uint8_t *const *seqs; const uint64_t *lens; int n;
    double seconds; int tid, nthreads;
    uint8_t *buf; snk_oracle_stream *st;
    uint64_t done, checksum; double elapsed;
} timed_job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *timed_worker(void *arg)
{
    timed_job *j = (timed_job *)arg;
    const double t0 = now_s();
    uint64_t k = (uint64_t)j->tid;
    for (;;) {
        const int a = (int)(k % (uint64_t)j->n), b = (int)((k * 7u + 1u + k / (uint64_t)j->n) % (uint64_t)j->n);
        const uint64_t nx = j->lens[a], ny = j->lens[b], n = nx + ny;
        memcpy(j->buf, j->seqs[a], nx);                 /* the concatenation the reference builds per pair */
        memcpy(j->buf + nx, j->seqs[b], ny);
        if (n > SNK_BLOCK) {
            snk_oracle_stream_init(j->st);
            snk_oracle_stream_run(j->st, j->buf, n, n, NULL);
            j->checksum += j->st->out + 4u;
        } else {
            j->checksum += snk_oracle_lz4f_size(j->buf, n);
        }
        j->done++;
        k += (uint64_t)j->nthreads;
        j->elapsed = now_s() - t0;
        if (j->elapsed >= j->seconds) break;
    }
    return NULL;
}

int snk_oracle_pairs_timed(const uint8_t *const *seqs, const uint64_t *lens, int n, int nthreads,
                           double seconds, uint64_t *pairs_done, double *elapsed, uint64_t *checksum)
{
    pthread_t *th;
    timed_job *jobs;
    uint64_t maxlen = 0;
    int t, rc = 0;
    if (n < 1 || nthreads < 1) return -1;
    for (t = 0; t < n; t++) if (lens[t] > maxlen) maxlen = lens[t];
    th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    jobs = (timed_job *)calloc((size_t)nthreads, sizeof(timed_job));
    if (!th || !jobs) return -1;
    for (t = 0; t < nthreads; t++) {
        jobs[t].seqs = seqs; jobs[t].lens = lens; jobs[t].n = n; jobs[t].seconds = seconds;
        jobs[t].tid = t; jobs[t].nthreads = nthreads;
        jobs[t].buf = (uint8_t *)malloc(2 * maxlen + 16);
        jobs[t].st = (snk_oracle_stream *)malloc(sizeof(snk_oracle_stream));
        if (!jobs[t].buf || !jobs[t].st) rc = -1;
        else memset(jobs[t].buf, 0, 2 * maxlen + 16);   /* touch the pages before the clock starts */
    }
    *pairs_done = 0; *elapsed = 0.0; *checksum = 0;
    if (rc == 0) {
        for (t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, timed_worker, &jobs[t]);
        for (t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        for (t = 0; t < nthreads; t++) {
            *pairs_done += jobs[t].done; *checksum += jobs[t].checksum;
            if (jobs[t].elapsed > *elapsed) *elapsed = jobs[t].elapsed;
        }
    }
    for (t = 0; t < nthreads; t++) { free(jobs[t].buf); free(jobs[t].st); }
    free(th); free(jobs);
    return rc;
}
