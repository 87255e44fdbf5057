/*
 * deflate_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or
 * called by the product (snacc_amd/); only tests/, __graft_entry__.smoke() and bench tooling
 * may use it, as the checker.
 *
 * CPU restatement of the arithmetic behind the reference's gzip and zlib paths:
 *
 *   ref:snacc/pairwise_ncd.py:73-74   gzip.compress(sequence)   -> zlib deflate level 9, 18 B wrapper
 *   ref:snacc/pairwise_ncd.py:77-78   zlib.compress(sequence)   -> zlib deflate level 6,  6 B wrapper
 *
 * The codec is a third-party dependency that is not part of /root/reference: CPython's
 * zlib/gzip modules over the system zlib (1.2.11 in this image; `zlib.ZLIB_RUNTIME_VERSION`).
 * This file restates zlib's published deflate algorithm (deflate_slow + longest_match + the
 * dynamic/static/stored block decision of trees.c) as a SIZE computation: no output bytes are
 * produced, only the exact number of bits of every block.  It keeps zlib's actual data
 * structures (64 KiB sliding window buffer, head/prev hash chains, 16 K symbol buffer) so that
 * every corner that depends on them (window slides, chain truncation, stale window bytes past the
 * end of the input, hash state at the end of the stream) comes out the same.
 *
 * Pinned by tests/test_deflate_oracle.py against the zlib binary itself (gzip.compress /
 * zlib.compress of the interpreter, differential fuzz + the reference's sample.fa sizes).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define W_SIZE 32768u
#define W_MASK (W_SIZE - 1u)
#define WINDOW_SIZE (2u * W_SIZE)
#define HASH_SIZE 32768u
#define HASH_MASK (HASH_SIZE - 1u)
#define HASH_SHIFT 5
#define MIN_MATCH 3
#define MAX_MATCH 258
#define MIN_LOOKAHEAD (MAX_MATCH + MIN_MATCH + 1)
#define MAX_DIST (W_SIZE - MIN_LOOKAHEAD)
#define TOO_FAR 4096
#define WIN_INIT MAX_MATCH
#define LIT_BUFSIZE 16384u

#define LITERALS 256
#define LENGTH_CODES 29
#define L_CODES (LITERALS + 1 + LENGTH_CODES)
#define D_CODES 30
#define BL_CODES 19
#define HEAP_SIZE (2 * L_CODES + 1)
#define END_BLOCK 256
#define REP_3_6 16
#define REPZ_3_10 17
#define REPZ_11_138 18

static const int extra_lbits[LENGTH_CODES] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const int extra_dbits[D_CODES] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const int extra_blbits[BL_CODES] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
static const uint8_t bl_order[BL_CODES] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

static uint8_t length_code[MAX_MATCH - MIN_MATCH + 1];
static uint8_t dist_code[512];
static uint8_t static_llen[L_CODES + 2];
static int tables_ready;

static void make_tables(void)
{
    int code, n, length = 0, dist = 0;
    if (tables_ready) return;
    for (code = 0; code < LENGTH_CODES - 1; code++)
        for (n = 0; n < (1 << extra_lbits[code]); n++) length_code[length++] = (uint8_t)code;
    length_code[length - 1] = (uint8_t)code;              /* match length 258 -> code 28 */
    for (code = 0; code < 16; code++)
        for (n = 0; n < (1 << extra_dbits[code]); n++) dist_code[dist++] = (uint8_t)code;
    dist >>= 7;
    for (; code < D_CODES; code++)
        for (n = 0; n < (1 << (extra_dbits[code] - 7)); n++) dist_code[256 + dist++] = (uint8_t)code;
    for (n = 0; n <= 143; n++) static_llen[n] = 8;
    for (; n <= 255; n++) static_llen[n] = 9;
    for (; n <= 279; n++) static_llen[n] = 7;
    for (; n <= 287; n++) static_llen[n] = 8;
    tables_ready = 1;
}

#define D_CODE(dist) ((dist) < 256 ? dist_code[dist] : dist_code[256 + ((dist) >> 7)])

typedef struct {
    unsigned good_length, max_lazy, nice_length, max_chain;
} dfl_config;

/* zlib's configuration_table, the deflate_slow rows (levels 4..9) */
static const dfl_config config_table[10] = {
    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
    {4, 4, 16, 16}, {8, 16, 32, 32}, {8, 16, 128, 128}, {8, 32, 128, 256}, {32, 128, 258, 1024}, {32, 258, 258, 4096}};

/* one Huffman tree under construction */
typedef struct {
    uint16_t freq[HEAP_SIZE];
    uint16_t dad[HEAP_SIZE];
    uint16_t len[HEAP_SIZE];
} dfl_tree;

typedef struct {
    /* input: the virtual concatenation a + b */
    const uint8_t *a, *b;
    size_t na, nb, in_pos;
    /* zlib's deflate_state, the part that decides sizes */
    uint8_t window[WINDOW_SIZE + 8];
    uint16_t prev[W_SIZE];
    uint16_t head[HASH_SIZE];
    unsigned ins_h, strstart, lookahead, match_start, match_length, prev_length, prev_match, insert;
    int match_available;
    long block_start;
    unsigned long high_water;
    dfl_config cfg;
    /* symbol statistics of the current block */
    dfl_tree lt, dt, bt;
    unsigned last_lit;
    long opt_len, static_len;
    int heap[HEAP_SIZE], heap_len, heap_max;
    uint8_t depth[HEAP_SIZE];
    size_t abs_pos;          /* stream position of strstart while priming */
    size_t start;            /* positions before `start` are only inserted into the hash chains (segment jobs) */
    /* result */
    uint64_t bits;
    /* optional trace of the emitted symbols / blocks (debug aid for the GPU port) */
    uint32_t *sym_trace; size_t sym_cap, sym_n;
    uint64_t *blk_trace; size_t blk_cap, blk_n;
    uint64_t *win_trace;     /* per block: loop-top stream position << 32 | stream position of window[0] at the flush */
    uint64_t looptop;
} dfl_state;

static size_t read_buf(dfl_state *s, uint8_t *dst, size_t size)
{
    const size_t total = s->na + s->nb;
    size_t len = total - s->in_pos, i;
    if (len > size) len = size;
    for (i = 0; i < len; i++) {
        const size_t p = s->in_pos + i;
        dst[i] = p < s->na ? s->a[p] : s->b[p - s->na];
    }
    s->in_pos += len;
    return len;
}

#define UPDATE_HASH(s, h, c) (h = (((h) << HASH_SHIFT) ^ (c)) & HASH_MASK)
#define INSERT_STRING(s, str, match_head) \
    (UPDATE_HASH(s, (s)->ins_h, (s)->window[(str) + (MIN_MATCH - 1)]), \
     match_head = (s)->prev[(str) & W_MASK] = (s)->head[(s)->ins_h], \
     (s)->head[(s)->ins_h] = (uint16_t)(str))

static void slide_hash(dfl_state *s)
{
    unsigned n;
    for (n = 0; n < HASH_SIZE; n++) s->head[n] = (uint16_t)(s->head[n] >= W_SIZE ? s->head[n] - W_SIZE : 0);
    for (n = 0; n < W_SIZE; n++) s->prev[n] = (uint16_t)(s->prev[n] >= W_SIZE ? s->prev[n] - W_SIZE : 0);
}

static void fill_window(dfl_state *s)
{
    unsigned n, more;
    const size_t total = s->na + s->nb;
    do {
        more = (unsigned)(WINDOW_SIZE - s->lookahead - s->strstart);
        if (s->strstart >= W_SIZE + MAX_DIST) {
            memcpy(s->window, s->window + W_SIZE, W_SIZE - more);
            s->match_start -= W_SIZE;
            s->strstart -= W_SIZE;
            s->block_start -= (long)W_SIZE;
            slide_hash(s);
            more += W_SIZE;
        }
        if (s->in_pos == total) break;
        n = (unsigned)read_buf(s, s->window + s->strstart + s->lookahead, more);
        s->lookahead += n;
        if (s->lookahead + s->insert >= MIN_MATCH) {
            unsigned str = s->strstart - s->insert;
            s->ins_h = s->window[str];
            UPDATE_HASH(s, s->ins_h, s->window[str + 1]);
            while (s->insert) {
                UPDATE_HASH(s, s->ins_h, s->window[str + MIN_MATCH - 1]);
                s->prev[str & W_MASK] = s->head[s->ins_h];
                s->head[s->ins_h] = (uint16_t)str;
                str++;
                s->insert--;
                if (s->lookahead + s->insert < MIN_MATCH) break;
            }
        }
    } while (s->lookahead < MIN_LOOKAHEAD && s->in_pos != total);

    if (s->high_water < WINDOW_SIZE) {
        unsigned long curr = s->strstart + (unsigned long)s->lookahead, init;
        if (s->high_water < curr) {
            init = WINDOW_SIZE - curr;
            if (init > WIN_INIT) init = WIN_INIT;
            memset(s->window + curr, 0, init);
            s->high_water = curr + init;
        } else if (s->high_water < curr + WIN_INIT) {
            init = curr + WIN_INIT - s->high_water;
            if (init > WINDOW_SIZE - s->high_water) init = WINDOW_SIZE - s->high_water;
            memset(s->window + s->high_water, 0, init);
            s->high_water += init;
        }
    }
}

static unsigned longest_match(dfl_state *s, unsigned cur_match)
{
    unsigned chain_length = s->cfg.max_chain;
    const uint8_t *scan = s->window + s->strstart;
    int best_len = (int)s->prev_length;
    int nice_match = (int)s->cfg.nice_length;
    const unsigned limit = s->strstart > MAX_DIST ? s->strstart - MAX_DIST : 0;
    uint8_t scan_end1 = scan[best_len - 1], scan_end = scan[best_len];

    if (s->prev_length >= s->cfg.good_length) chain_length >>= 2;
    if ((unsigned)nice_match > s->lookahead) nice_match = (int)s->lookahead;
    do {
        const uint8_t *match = s->window + cur_match;
        int len;
        if (match[best_len] != scan_end || match[best_len - 1] != scan_end1 || match[0] != scan[0] || match[1] != scan[1])
            continue;
        /* byte 2 is taken as equal (same hash chain), exactly as zlib does */
        len = 3;
        while (len < MAX_MATCH && scan[len] == match[len]) len++;
        if (len > best_len) {
            s->match_start = cur_match;
            best_len = len;
            if (len >= nice_match) break;
            scan_end1 = scan[best_len - 1];
            scan_end = scan[best_len];
        }
    } while ((cur_match = s->prev[cur_match & W_MASK]) > limit && --chain_length != 0);
    return (unsigned)best_len <= s->lookahead ? (unsigned)best_len : s->lookahead;
}

/* ---------------------------------------------------------------- trees.c, sizes only */

static void init_block(dfl_state *s)
{
    memset(s->lt.freq, 0, sizeof s->lt.freq);
    memset(s->dt.freq, 0, sizeof s->dt.freq);
    memset(s->bt.freq, 0, sizeof s->bt.freq);
    s->lt.freq[END_BLOCK] = 1;
    s->opt_len = s->static_len = 0;
    s->last_lit = 0;
}

#define SMALLER(t, n, m) ((t)->freq[n] < (t)->freq[m] || ((t)->freq[n] == (t)->freq[m] && s->depth[n] <= s->depth[m]))

static void pqdownheap(dfl_state *s, dfl_tree *t, int k)
{
    const int v = s->heap[k];
    int j = k << 1;
    while (j <= s->heap_len) {
        if (j < s->heap_len && SMALLER(t, s->heap[j + 1], s->heap[j])) j++;
        if (SMALLER(t, v, s->heap[j])) break;
        s->heap[k] = s->heap[j];
        k = j;
        j <<= 1;
    }
    s->heap[k] = v;
}

/* build_tree + gen_bitlen.  `slen` = static code lengths (NULL for the bit-length tree). */
static int build_tree(dfl_state *s, dfl_tree *t, int elems, const int *extra, int extra_base, int max_length,
                      const uint8_t *slen, int slen_const)
{
    int n, m, max_code = -1, node, h, bits, overflow = 0;
    uint16_t bl_count[16];

    s->heap_len = 0;
    s->heap_max = HEAP_SIZE;
    for (n = 0; n < elems; n++) {
        if (t->freq[n] != 0) { s->heap[++s->heap_len] = max_code = n; s->depth[n] = 0; }
        else t->len[n] = 0;
    }
    while (s->heap_len < 2) {
        node = s->heap[++s->heap_len] = (max_code < 2 ? ++max_code : 0);
        t->freq[node] = 1;
        s->depth[node] = 0;
        s->opt_len--;
        if (slen || slen_const) s->static_len -= slen ? slen[node] : slen_const;
    }
    for (n = s->heap_len / 2; n >= 1; n--) pqdownheap(s, t, n);
    node = elems;
    do {
        n = s->heap[1];
        s->heap[1] = s->heap[s->heap_len--];
        pqdownheap(s, t, 1);
        m = s->heap[1];
        s->heap[--s->heap_max] = n;
        s->heap[--s->heap_max] = m;
        t->freq[node] = (uint16_t)(t->freq[n] + t->freq[m]);
        s->depth[node] = (uint8_t)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
        t->dad[n] = t->dad[m] = (uint16_t)node;
        s->heap[1] = node++;
        pqdownheap(s, t, 1);
    } while (s->heap_len >= 2);
    s->heap[--s->heap_max] = s->heap[1];

    /* gen_bitlen */
    memset(bl_count, 0, sizeof bl_count);
    t->len[s->heap[s->heap_max]] = 0;
    for (h = s->heap_max + 1; h < HEAP_SIZE; h++) {
        int xbits = 0;
        n = s->heap[h];
        bits = t->len[t->dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        t->len[n] = (uint16_t)bits;
        if (n > max_code) continue;
        bl_count[bits]++;
        if (n >= extra_base) xbits = extra[n - extra_base];
        s->opt_len += (long)t->freq[n] * (bits + xbits);
        if (slen || slen_const) s->static_len += (long)t->freq[n] * ((slen ? slen[n] : slen_const) + xbits);
    }
    if (overflow > 0) {
        do {
            bits = max_length - 1;
            while (bl_count[bits] == 0) bits--;
            bl_count[bits]--;
            bl_count[bits + 1] += 2;
            bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (bits = max_length; bits != 0; bits--) {
            n = bl_count[bits];
            while (n != 0) {
                m = s->heap[--h];
                if (m > max_code) continue;
                if (t->len[m] != (unsigned)bits) {
                    s->opt_len += ((long)bits - (long)t->len[m]) * (long)t->freq[m];
                    t->len[m] = (uint16_t)bits;
                }
                n--;
            }
        }
    }
    return max_code;
}

static void scan_tree(dfl_state *s, dfl_tree *t, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = t->len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    t->len[max_code + 1] = 0xffff;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen;
        nextlen = t->len[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) s->bt.freq[curlen] = (uint16_t)(s->bt.freq[curlen] + count);
        else if (curlen != 0) {
            if (curlen != prevlen) s->bt.freq[curlen]++;
            s->bt.freq[REP_3_6]++;
        } else if (count <= 10) s->bt.freq[REPZ_3_10]++;
        else s->bt.freq[REPZ_11_138]++;
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

static void flush_block(dfl_state *s, int have_buf, unsigned long stored_len, int last)
{
    unsigned long opt_lenb, static_lenb;
    int lmax, dmax, max_blindex;
    uint64_t before = s->bits;

    lmax = build_tree(s, &s->lt, L_CODES, extra_lbits, LITERALS + 1, 15, static_llen, 0);
    dmax = build_tree(s, &s->dt, D_CODES, extra_dbits, 0, 15, NULL, 5);
    scan_tree(s, &s->lt, lmax);
    scan_tree(s, &s->dt, dmax);
    build_tree(s, &s->bt, BL_CODES, extra_blbits, 0, 7, NULL, 0);
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
        if (s->bt.len[bl_order[max_blindex]] != 0) break;
    s->opt_len += 3 * ((long)max_blindex + 1) + 5 + 5 + 4;

    opt_lenb = (unsigned long)(s->opt_len + 3 + 7) >> 3;
    static_lenb = (unsigned long)(s->static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;

    if (stored_len + 4 <= opt_lenb && have_buf) {
        s->bits += 3;
        s->bits = (s->bits + 7) & ~(uint64_t)7;
        s->bits += 32 + 8 * (uint64_t)stored_len;
    } else if (static_lenb == opt_lenb) {
        s->bits += 3 + (uint64_t)s->static_len;
    } else {
        s->bits += 3 + (uint64_t)s->opt_len;
    }
    if (s->blk_trace && s->blk_n < s->blk_cap) s->blk_trace[s->blk_n] = s->bits - before;
    if (s->win_trace && s->blk_n < s->blk_cap)
        s->win_trace[s->blk_n] = (s->looptop << 32) | (uint64_t)(s->in_pos - s->lookahead - s->strstart);
    s->blk_n++;
    init_block(s);
    if (last) s->bits = (s->bits + 7) & ~(uint64_t)7;
}

#define FLUSH_BLOCK(s, last) do { \
    flush_block(s, (s)->block_start >= 0L, (unsigned long)((long)(s)->strstart - (s)->block_start), (last)); \
    (s)->block_start = (long)(s)->strstart; } while (0)

static int tally_lit(dfl_state *s, unsigned c)
{
    if (s->sym_trace && s->sym_n < s->sym_cap) s->sym_trace[s->sym_n] = c;
    s->sym_n++;
    s->last_lit++;
    s->lt.freq[c]++;
    return s->last_lit == LIT_BUFSIZE - 1;
}

static int tally_dist(dfl_state *s, unsigned dist, unsigned lc)
{
    if (s->sym_trace && s->sym_n < s->sym_cap) s->sym_trace[s->sym_n] = 0x80000000u | (lc << 16) | dist;
    s->sym_n++;
    s->last_lit++;
    dist--;
    s->lt.freq[length_code[lc] + LITERALS + 1]++;
    s->dt.freq[D_CODE(dist)]++;
    return s->last_lit == LIT_BUFSIZE - 1;
}

/* deflate_slow over the whole input, ending with Z_FINISH */
static void deflate_slow(dfl_state *s)
{
    unsigned hash_head;
    int bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead == 0) break;
        }
        s->looptop = (uint64_t)(s->in_pos - s->lookahead);      /* stream position of strstart at this loop top */
        hash_head = 0;
        if (s->lookahead >= MIN_MATCH) INSERT_STRING(s, s->strstart, hash_head);
        if (s->abs_pos < s->start) {                 /* priming: like a preset dictionary, nothing is parsed or emitted */
            s->strstart++;
            s->lookahead--;
            s->abs_pos++;
            s->block_start = (long)s->strstart;
            continue;
        }
        s->prev_length = s->match_length;
        s->prev_match = s->match_start;
        s->match_length = MIN_MATCH - 1;
        if (hash_head != 0 && s->prev_length < s->cfg.max_lazy && s->strstart - hash_head <= MAX_DIST) {
            s->match_length = longest_match(s, hash_head);
            if (s->match_length <= 5 && s->match_length == MIN_MATCH && s->strstart - s->match_start > TOO_FAR)
                s->match_length = MIN_MATCH - 1;
        }
        if (s->prev_length >= MIN_MATCH && s->match_length <= s->prev_length) {
            const unsigned max_insert = s->strstart + s->lookahead - MIN_MATCH;
            bflush = tally_dist(s, s->strstart - 1 - s->prev_match, s->prev_length - MIN_MATCH);
            s->lookahead -= s->prev_length - 1;
            s->prev_length -= 2;
            do {
                if (++s->strstart <= max_insert) INSERT_STRING(s, s->strstart, hash_head);
            } while (--s->prev_length != 0);
            s->match_available = 0;
            s->match_length = MIN_MATCH - 1;
            s->strstart++;
            if (bflush) FLUSH_BLOCK(s, 0);
        } else if (s->match_available) {
            bflush = tally_lit(s, s->window[s->strstart - 1]);
            if (bflush) FLUSH_BLOCK(s, 0);
            s->strstart++;
            s->lookahead--;
        } else {
            s->match_available = 1;
            s->strstart++;
            s->lookahead--;
        }
    }
    if (s->match_available) {
        (void)tally_lit(s, s->window[s->strstart - 1]);
        s->match_available = 0;
    }
    s->looptop = (uint64_t)s->in_pos;
    FLUSH_BLOCK(s, 1);
}

static dfl_state *dfl_new(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, int level)
{
    dfl_state *s;
    make_tables();
    if (level < 4 || level > 9) return NULL;
    s = (dfl_state *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->a = a; s->na = na; s->b = b; s->nb = b ? nb : 0;
    s->cfg = config_table[level];
    s->match_length = s->prev_length = MIN_MATCH - 1;
    init_block(s);
    return s;
}

/* Length in BYTES of the raw deflate stream zlib (1.2.11, memLevel 8, windowBits 15, default
 * strategy) produces for the concatenation a+b at `level` (4..9).  0 on bad arguments. */
uint64_t dfl_oracle_raw_size(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, int level)
{
    uint64_t r;
    dfl_state *s = dfl_new(a, na, b, nb, level);
    if (!s) return 0;
    deflate_slow(s);
    r = s->bits >> 3;
    free(s);
    return r;
}

/* Same, also recording the symbol stream (literal byte, or 0x80000000 | (len-3) << 16 | dist)
 * and the bit count of every block.  Returns the raw size; *n_sym / *n_blk receive the totals. */
uint64_t dfl_oracle_trace(const uint8_t *a, size_t na, const uint8_t *b, size_t nb, int level,
                          uint32_t *sym, size_t sym_cap, size_t *n_sym, uint64_t *blk, size_t blk_cap, size_t *n_blk)
{
    uint64_t r;
    dfl_state *s = dfl_new(a, na, b, nb, level);
    if (!s) return 0;
    s->sym_trace = sym; s->sym_cap = sym_cap;
    s->blk_trace = blk; s->blk_cap = blk_cap;
    deflate_slow(s);
    r = s->bits >> 3;
    if (n_sym) *n_sym = s->sym_n;
    if (n_blk) *n_blk = s->blk_n;
    free(s);
    return r;
}

/* Block cost alone (the trees.c part): frequencies of the 286 literal/length codes (END_BLOCK
 * included by the caller or not -- it is forced to 1 as zlib does) and of the 30 distance codes
 * -> bits the block adds to the stream when it is not stored, i.e. min(static, dynamic) + 3. */
/* The symbol stream of a parser that starts at stream position `start` "as if right behind a match", with every
 * earlier position in the hash chains but nothing parsed before (what a segment job of the GPU path does; zlib
 * itself behaves like this after deflateSetDictionary).  Symbols as in dfl_oracle_trace. */
uint64_t dfl_oracle_trace_from(const uint8_t *a, size_t na, size_t start, int level, uint32_t *sym, size_t sym_cap, size_t *n_sym)
{
    uint64_t r;
    dfl_state *s = dfl_new(a, na, NULL, 0, level);
    if (!s || start > na) { free(s); return 0; }
    s->start = start;
    s->sym_trace = sym; s->sym_cap = sym_cap;
    deflate_slow(s);
    r = s->bits >> 3;
    if (n_sym) *n_sym = s->sym_n;
    free(s);
    return r;
}

/* What zlib's compare loop can read behind the end of the input: the 258 window bytes that follow the last input
 * byte, as they stand when the stream is finished. */
void dfl_oracle_bytes_behind_end(const uint8_t *a, size_t na, int level, uint8_t *out /* [258] */)
{
    dfl_state *s = dfl_new(a, na, NULL, 0, level);
    if (!s) return;
    deflate_slow(s);
    for (unsigned i = 0; i < MAX_MATCH; i++) {
        const unsigned idx = s->strstart + s->lookahead + i;
        out[i] = idx < WINDOW_SIZE ? s->window[idx] : 0;
    }
    free(s);
}

/* Per block: (loop-top stream position << 32) | stream position of window[0] when the block was flushed.
 * Returns the number of blocks. */
size_t dfl_oracle_window_trace(const uint8_t *a, size_t na, int level, uint64_t *win, size_t cap)
{
    size_t r;
    dfl_state *s = dfl_new(a, na, NULL, 0, level);
    if (!s) return 0;
    s->win_trace = win; s->blk_cap = cap;
    deflate_slow(s);
    r = s->blk_n;
    free(s);
    return r;
}

uint64_t dfl_oracle_block_bits(const uint16_t *lfreq, const uint16_t *dfreq)
{
    uint64_t r;
    dfl_state *s = dfl_new(NULL, 0, NULL, 0, 9);
    if (!s) return 0;
    memcpy(s->lt.freq, lfreq, L_CODES * sizeof(uint16_t));
    memcpy(s->dt.freq, dfreq, D_CODES * sizeof(uint16_t));
    s->lt.freq[END_BLOCK] = 1;
    flush_block(s, 0, 0, 0);
    r = s->bits;
    free(s);
    return r;
}
