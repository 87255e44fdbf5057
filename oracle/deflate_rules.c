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
