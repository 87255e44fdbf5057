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

uint64_t dfl_rules_raw_size(const uint8_t *a, size_t n, int level)
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
            unsigned chain = s->cfg.max_chain, nice = s->cfg.nice_length, best = prev_length, j = 0;
            int searched = 0;
            int32_t q = prevpos[p];
            if (prev_length >= s->cfg.good_length) chain >>= 2;
            if (nice > la) nice = (unsigned)la;
            while (q >= 0 && j < chain) {
                const size_t dist = p - (size_t)q;
                unsigned len = 0;
                if (q == 0 || dist > (j == 0 ? 32506u : 32505u)) break;
                searched = 1;
                while (len < MAX_MATCH && rules_byte(&S, p + len) == rules_byte(&S, (size_t)q + len)) len++;
                if (len > best) {
                    best = len;
                    match_start = (unsigned)q;
                    if (len >= nice) break;
                }
                q = prevpos[q];
                j++;
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
