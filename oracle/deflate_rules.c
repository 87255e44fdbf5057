/*
 * deflate_rules.c -- TEST INFRASTRUCTURE ONLY.  The RULES the GPU kernel applies (snacc_amd/csrc/snk_deflate.hip.h,
 * DESIGN.md section 10), executable on the CPU, so that they can be checked against the window-faithful
 * restatement in deflate_oracle.c (and through it against zlib) without a GPU:
 *
 *  - the hash chain of a position is data only: the earlier positions (<= n - 3) with the same 15-bit hash of three
 *    bytes, most recent first -- no window buffer, no head/prev arrays, no slides;
 *  - a chain walk looks at up to max_chain members (a quarter after a good match), stops at the first that is out
 *    of range (the head may be 32 506 back, every later member 32 505, stream position 0 never counts), keeps the
 *    first longest, and ends early at nice_match (clamped to the bytes left);
 *  - bytes behind the end of the input read as zeros until zlib's window has slid once, then as the bytes 32 KiB
 *    earlier; the window's position is the closed form of dfl_window_base;
 *  - lazy evaluation, TOO_FAR, blocks of 16 383 symbols and their pricing exactly as zlib (the pricing code is
 *    shared with deflate_oracle.c).
 *
 * The work elision of the GPU path (restart from x's stream, resynchronisation with y's, segments) is not modelled
 * here; tests/test_deflate_oracle.py checks those arguments on the symbol streams.
 */
#include "deflate_oracle.c"

static unsigned rules_window_base(unsigned p0, size_t n)
{
    unsigned base = p0 >= 65275u ? ((p0 - 65275u) >> 15) << 15 : 0u;
    for (;;) {
        const unsigned t = base + (n <= (size_t)base + 65535u ? 65274u : 65275u);
        if (p0 < t) break;
        base += 32768u;
    }
    return base;
}

typedef struct { const uint8_t *a; size_t n; int slid; } rules_stream;

static unsigned rules_byte(const rules_stream *s, size_t i)
{
    if (i >= s->n) {
        if (!s->slid) return 0;
        i -= 32768;
    }
    return s->a[i];
}

/* One chain walk by the rules.  Returns 1 if a search took place; *best / *mstart as zlib leaves them. */
static int rules_walk(const rules_stream *S, const int32_t *prevpos, size_t p, unsigned prev_length, unsigned chain,
                      unsigned nice, unsigned *best_out, unsigned *mstart)
{
    unsigned best = prev_length, j = 0;
    int searched = 0;
    int32_t q = prevpos[p];
    while (q >= 0 && j < chain) {
        const size_t dist = p - (size_t)q;
        unsigned len = 0;
        if (q == 0 || dist > (j == 0 ? 32506u : 32505u)) break;
        searched = 1;
        while (len < MAX_MATCH && rules_byte(S, p + len) == rules_byte(S, (size_t)q + len)) len++;
        if (len > best) {
            best = len;
            *mstart = (unsigned)q;
            if (len >= nice) break;
        }
        q = prevpos[q];
        j++;
    }
    *best_out = best;
    return searched;
}

static unsigned rules_hash6(const uint8_t *p)
{
    uint64_t v = 0;
    memcpy(&v, p, 6);
    return (unsigned)((v * 0x9E3779B97F4A7C15ull) >> 48);
}

/* The K-pass of the gzip kernel: only the chain members that share the probe's 16-bit hash of six bytes (and its
 * 3-byte hash), each with its position j3 in the full chain.  Returns 1 if the kernel would take the result
 * ("a match of >= 6 bytes that beats prev_length"), 2 if it would skip the search because prev_length >= 5. */
static int rules_kpass(const rules_stream *S, const int32_t *prevpos, size_t p, unsigned prev_length, unsigned chain,
                       unsigned nice, unsigned *best_out, unsigned *mstart)
{
    const unsigned h6 = rules_hash6(S->a + p);
    unsigned best = prev_length, j3 = 0, ms = *mstart;
    int32_t q = prevpos[p];
    for (; q >= 0; q = prevpos[q], j3++) {
        const size_t dist = p - (size_t)q;
        unsigned len = 0;
        if (dist > (j3 == 0 ? 32506u : 32505u)) break;          /* the lists are in falling position order */
        if ((size_t)q + 6 > S->n || rules_hash6(S->a + q) != h6) continue;   /* not in the six-byte bucket */
        if (q == 0 || j3 >= chain) continue;
        while (len < MAX_MATCH && rules_byte(S, p + len) == rules_byte(S, (size_t)q + len)) len++;
        if (len > best) {
            best = len;
            ms = (unsigned)q;
            if (len >= nice) break;
        }
    }
    if (best >= 6 && best > prev_length) { *best_out = best; *mstart = ms; return 1; }
    return prev_length >= 5 ? 2 : 0;
}

/* Runs the parse by the rules and counts the probes at which the K-pass would have decided differently from the full
 * chain walk (0 = the argument of DESIGN.md section 10 holds on this input).  *n_k = probes the K-pass decided. */
uint64_t dfl_rules_check_kpass(const uint8_t *a, size_t n, int level, uint64_t *n_k);

uint64_t dfl_rules_raw_size_impl(const uint8_t *a, size_t n, int level, uint64_t *violations, uint64_t *n_k);

uint64_t dfl_rules_raw_size(const uint8_t *a, size_t n, int level)
{
    return dfl_rules_raw_size_impl(a, n, level, NULL, NULL);
}

uint64_t dfl_rules_check_kpass(const uint8_t *a, size_t n, int level, uint64_t *n_k)
{
    uint64_t v = 0;
    (void)dfl_rules_raw_size_impl(a, n, level, &v, n_k);
    return v;
}

uint64_t dfl_rules_raw_size_impl(const uint8_t *a, size_t n, int level, uint64_t *violations, uint64_t *n_k)
{
    dfl_state *s = dfl_new(a, n, NULL, 0, level);
    int32_t *prevpos, *last;
    rules_stream S;
    size_t p = 0, i;
    unsigned match_length = 2, match_start = 0;
    int match_available = 0;
    long block_start = 0;
    uint64_t r;
    if (!s) return 0;
    prevpos = (int32_t *)malloc((n + 1) * sizeof(int32_t));
    last = (int32_t *)malloc(HASH_SIZE * sizeof(int32_t));
    for (i = 0; i < HASH_SIZE; i++) last[i] = -1;
    for (i = 0; i + 3 <= n; i++) {                              /* every position with three bytes left is in its chain */
        const unsigned h = (((unsigned)a[i] << 10) ^ ((unsigned)a[i + 1] << 5) ^ a[i + 2]) & HASH_MASK;
        prevpos[i] = last[h];
        last[h] = (int32_t)i;
    }
    S.a = a; S.n = n; S.slid = 0;

    while (p < n) {
        const size_t la = n - p;
        const unsigned prev_length = match_length, prev_match = match_start;
        int flush = 0;
        unsigned end = 0;
        S.slid = n > 65536 || p >= (n <= 65535 ? 65274u : 65275u);
        match_length = 2;
        if (la >= 3 && prev_length < s->cfg.max_lazy) {
            unsigned chain = s->cfg.max_chain, nice = s->cfg.nice_length, best = prev_length;
            int searched;
            const unsigned ms_in = match_start;
            if (prev_length >= s->cfg.good_length) chain >>= 2;
            if (nice > la) nice = (unsigned)la;
            searched = rules_walk(&S, prevpos, p, prev_length, chain, nice, &best, &match_start);
            if (violations && la >= MIN_LOOKAHEAD) {              /* where the kernel's K-pass applies */
                unsigned kbest = prev_length, kms = ms_in;
                const int k = rules_kpass(&S, prevpos, p, prev_length, chain, nice, &kbest, &kms);
                if (k == 1) { if (n_k) (*n_k)++; if (!searched || kbest != best || kms != match_start) (*violations)++; }
                else if (k == 2) {                                /* the emission must not depend on this search */
                    const unsigned ml = searched ? (best <= la ? best : (unsigned)la) : 2u;
                    if (n_k) (*n_k)++;
                    if (ml > prev_length) (*violations)++;
                }
            }
            if (searched) {
                match_length = best <= la ? best : (unsigned)la;
                if (match_length == 3 && p - match_start > TOO_FAR) match_length = 2;
            }
        }
        if (prev_length >= 3 && match_length <= prev_length) {
            flush = tally_dist(s, (unsigned)(p - 1 - prev_match), prev_length - MIN_MATCH);
            p = p - 1 + prev_length;
            end = (unsigned)p;
            match_available = 0;
            match_length = 2;
            if (flush) {
                flush_block(s, block_start >= (long)rules_window_base(end - prev_length + 1, n), (unsigned long)(end - block_start), 0);
                block_start = (long)end;
            }
        } else if (match_available) {
            flush = tally_lit(s, a[p - 1]);
            if (flush) {
                flush_block(s, block_start >= (long)rules_window_base((unsigned)p, n), (unsigned long)(p - block_start), 0);
                block_start = (long)p;
            }
            p++;
        } else {
            match_available = 1;
            p++;
        }
    }
    if (match_available) (void)tally_lit(s, a[n - 1]);
    flush_block(s, block_start >= (long)rules_window_base((unsigned)n, n), (unsigned long)(n - block_start), 1);
    r = s->bits >> 3;
    free(prevpos); free(last); free(s);
    return r;
}

/* ------------------------------------------------------------------------------------------------------------
 * The PAIR job of the GPU path (point 2), as a CPU program: chains per sequence plus the two
 * seam positions whose three bytes mix x and y; restart from x's own stream ~600 bytes before the seam; parse until
 * the pair's parser and y's own parser stand right behind a match at the same position (>= 32 507 bytes after the
 * seam), continue with y's stored symbols; price the spliced stream.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct { uint32_t *sym, *pos; size_t n; } rules_syms;       /* symbols as dfl_oracle_trace, and where they start */

typedef struct {
    const uint8_t *x, *y;
    size_t lx, ly, n;
    const int32_t *px, *py, *lastx;       /* chains of x and of y; most recent position of every hash in x */
    int slid;
} rules_pair;

static unsigned pair_byte(const rules_pair *P, size_t i)
{
    if (i >= P->n) {
        if (!P->slid) return 0;
        i -= 32768;
    }
    return i < P->lx ? P->x[i] : P->y[i - P->lx];
}

static unsigned pair_hash3(const rules_pair *P, size_t i)
{
    return ((pair_byte(P, i) << 10) ^ (pair_byte(P, i + 1) << 5) ^ pair_byte(P, i + 2)) & HASH_MASK;
}

static int32_t *rules_chains(const uint8_t *a, size_t n, int32_t *last /* HASH_SIZE, may be NULL */)
{
    int32_t *prev = (int32_t *)malloc((n + 1) * sizeof(int32_t)), *l = last ? last : (int32_t *)malloc(HASH_SIZE * sizeof(int32_t));
    size_t i;
    for (i = 0; i < HASH_SIZE; i++) l[i] = -1;
    for (i = 0; i + 3 <= n; i++) {
        const unsigned h = (((unsigned)a[i] << 10) ^ ((unsigned)a[i + 1] << 5) ^ a[i + 2]) & HASH_MASK;
        prev[i] = l[h];
        l[h] = (int32_t)i;
    }
    if (!last) free(l);
    return prev;
}

/* The chain of stream position p, member after member: state machine over "y's chain, the seam positions, x's chain". */
typedef struct { int stage; int32_t q; unsigned h; } pair_iter;

static long pair_next(const rules_pair *P, size_t p, pair_iter *it)
{
    for (;;) {
        if (it->stage == 0) {                                   /* inside y */
            it->stage = 1;
            if (p >= P->lx) { it->q = P->py[p - P->lx]; } else { it->stage = 4; it->q = p + 3 <= P->lx ? P->px[p] : -2; }
            continue;
        }
        if (it->stage == 1) {                                   /* y's chain */
            if (it->q >= 0) { const long r = (long)P->lx + it->q; it->q = P->py[it->q]; return r; }
            it->stage = 2;
            continue;
        }
        if (it->stage == 2) {                                   /* seam position lx - 1 (three bytes: 1 of x, 2 of y) */
            it->stage = 3;
            if (P->lx >= 1 && P->lx - 1 + 3 <= P->n && P->lx - 1 < p && pair_hash3(P, P->lx - 1) == it->h) return (long)P->lx - 1;
            continue;
        }
        if (it->stage == 3) {                                   /* seam position lx - 2 */
            it->stage = 4;
            it->q = P->lastx[it->h];
            if (P->lx >= 2 && P->lx - 2 + 3 <= P->n && P->lx - 2 < p && pair_hash3(P, P->lx - 2) == it->h) return (long)P->lx - 2;
            continue;
        }
        /* stage 4: x's chain; q == -2 means "p is one of the two seam positions: start from the seam / x's top" */
        if (it->q == -2) { it->stage = (p == P->lx - 1) ? 3 : 4; it->q = P->lastx[it->h]; if (it->stage == 3) continue; }
        if (it->q >= 0) { const long r = it->q; it->q = P->px[it->q]; return r; }
        return -1;
    }
}

/* Parse the stream x+y from the clean state at `start` until `stop_sync` says so or the end; symbols are appended. */
static size_t pair_parse(rules_pair *P, const dfl_config *cfg, size_t start, const rules_syms *sy, rules_syms *out, size_t *sync_k)
{
    size_t p = start;
    unsigned match_length = 2, match_start = 0;
    int match_available = 0;
    const size_t n = P->n;
    *sync_k = (size_t)-1;
    while (p < n) {
        const size_t la = n - p;
        const unsigned prev_length = match_length, prev_match = match_start;
        P->slid = n > 65536 || p >= (n <= 65535 ? 65274u : 65275u);
        match_length = 2;
        if (la >= 3 && prev_length < cfg->max_lazy) {
            unsigned chain = cfg->max_chain, nice = cfg->nice_length, best = prev_length, j = 0;
            int searched = 0;
            pair_iter it = {0, -1, pair_hash3(P, p)};
            if (prev_length >= cfg->good_length) chain >>= 2;
            if (nice > la) nice = (unsigned)la;
            while (j < chain) {
                const long q = pair_next(P, p, &it);
                unsigned len = 0;
                if (q < 0) break;
                if (q == 0 || p - (size_t)q > (j == 0 ? 32506u : 32505u)) break;
                searched = 1;
                while (len < MAX_MATCH && pair_byte(P, p + len) == pair_byte(P, (size_t)q + len)) len++;
                if (len > best) { best = len; match_start = (unsigned)q; if (len >= nice) break; }
                j++;
            }
            if (searched) {
                match_length = best <= la ? best : (unsigned)la;
                if (match_length == 3 && p - match_start > TOO_FAR) match_length = 2;
            }
        }
        if (prev_length >= 3 && match_length <= prev_length) {
            out->sym[out->n] = 0x80000000u | ((prev_length - 3) << 16) | (unsigned)(p - 1 - prev_match);
            out->pos[out->n++] = (uint32_t)(p - 1);
            p = p - 1 + prev_length;
            match_available = 0;
            match_length = 2;
            if (sy && p >= P->lx + 32507 && p < n && P->ly > 65536) {    /* both right behind a match at the same place? */
                size_t lo = 0, hi = sy->n;
                const size_t want = p - P->lx;
                while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (sy->pos[mid] < want) lo = mid + 1; else hi = mid; }
                if (lo < sy->n && lo > 0 && sy->pos[lo] == want && (sy->sym[lo - 1] >> 31)) { *sync_k = lo; return p; }
            }
        } else if (match_available) {
            out->sym[out->n] = pair_byte(P, p - 1);
            out->pos[out->n++] = (uint32_t)(p - 1);
            p++;
        } else {
            match_available = 1;
            p++;
        }
    }
    if (match_available) { out->sym[out->n] = pair_byte(P, n - 1); out->pos[out->n++] = (uint32_t)(n - 1); }
    return p;
}

static rules_syms rules_own_stream(const uint8_t *a, size_t n, const dfl_config *cfg)
{
    rules_pair P = {a, a, n, 0, n, NULL, NULL, NULL, 0};
    int32_t *last = (int32_t *)malloc(HASH_SIZE * sizeof(int32_t));
    int32_t *prev = rules_chains(a, n, last);
    rules_syms out = {(uint32_t *)malloc((n + 1) * 4), (uint32_t *)malloc((n + 1) * 4), 0};
    size_t k;
    P.px = prev; P.lastx = last;
    pair_parse(&P, cfg, 0, NULL, &out, &k);
    free(prev); free(last);
    return out;
}

/* Price a symbol stream of total length n: blocks of 16 383 symbols, the literal at n - 1 never closes one. */
static uint64_t rules_price(dfl_state *s, const rules_syms *st, size_t n)
{
    long block_start = 0;
    size_t k;
    for (k = 0; k < st->n; k++) {
        const uint32_t v = st->sym[k];
        const size_t q = st->pos[k];
        int flush;
        if (v >> 31) flush = tally_dist(s, v & 0xffffu, (v >> 16) & 0x7fffu); else flush = tally_lit(s, v);
        if (flush && !(!(v >> 31) && q + 1 == n)) {
            const size_t end = (v >> 31) ? q + ((v >> 16) & 0x7fffu) + 3 : q + 1;
            flush_block(s, block_start >= (long)rules_window_base((unsigned)q + 1, n), (unsigned long)(end - block_start), 0);
            block_start = (long)end;
        }
    }
    flush_block(s, block_start >= (long)rules_window_base((unsigned)n, n), (unsigned long)(n - block_start), 1);
    return s->bits >> 3;
}

uint64_t dfl_rules_pair_size(const uint8_t *x, size_t lx, const uint8_t *y, size_t ly, int level)
{
    dfl_state *s = dfl_new(x, lx, y, ly, level);
    rules_syms sx, sy, seam, all;
    rules_pair P;
    int32_t *lastx, *px, *py;
    size_t k0 = 0, start = 0, sync_k, k;
    uint64_t r;
    if (!s) return 0;
    sx = rules_own_stream(x, lx, &s->cfg);
    sy = rules_own_stream(y, ly, &s->cfg);
    lastx = (int32_t *)malloc(HASH_SIZE * sizeof(int32_t));
    px = rules_chains(x, lx, lastx);
    py = rules_chains(y, ly, NULL);
    P.x = x; P.y = y; P.lx = lx; P.ly = ly; P.n = lx + ly; P.px = px; P.py = py; P.lastx = lastx; P.slid = 0;
    /* restart: the last symbol boundary of x's stream that lies right behind a match and >= 600 bytes before the seam */
    if (lx > 600 && sx.n) {
        size_t lo = 0, hi = sx.n;
        while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (sx.pos[mid] <= lx - 600) lo = mid + 1; else hi = mid; }
        k0 = lo ? lo - 1 : 0;
        while (k0 > 0 && !(sx.sym[k0 - 1] >> 31)) k0--;
        start = k0 < sx.n ? sx.pos[k0] : 0;
        if (k0 >= sx.n) { k0 = 0; start = 0; }
    }
    seam.sym = (uint32_t *)malloc((P.n + 1) * 4); seam.pos = (uint32_t *)malloc((P.n + 1) * 4); seam.n = 0;
    pair_parse(&P, &s->cfg, start, &sy, &seam, &sync_k);
    all.sym = (uint32_t *)malloc((P.n + 1) * 4); all.pos = (uint32_t *)malloc((P.n + 1) * 4); all.n = 0;
    for (k = 0; k < k0; k++) { all.sym[all.n] = sx.sym[k]; all.pos[all.n++] = sx.pos[k]; }
    for (k = 0; k < seam.n; k++) { all.sym[all.n] = seam.sym[k]; all.pos[all.n++] = seam.pos[k]; }
    if (sync_k != (size_t)-1)
        for (k = sync_k; k < sy.n; k++) { all.sym[all.n] = sy.sym[k]; all.pos[all.n++] = (uint32_t)(sy.pos[k] + lx); }
    r = rules_price(s, &all, P.n);
    free(sx.sym); free(sx.pos); free(sy.sym); free(sy.pos); free(seam.sym); free(seam.pos); free(all.sym); free(all.pos);
    free(lastx); free(px); free(py); free(s);
    return r;
}
