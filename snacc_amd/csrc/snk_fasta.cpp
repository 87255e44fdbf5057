// snk_fasta.cpp -- host-side FASTA ingest for the lz4 NCD path.
//
// Replaces, for the batched lz4 path, the per-task Biopython parse of
// ref:snacc/pairwise_ncd.py:29-39 (every file parsed 2N+1 times under the GIL) with one
// multi-threaded pass.  Semantics restated from Biopython's plain FASTA reader as used by the
// reference (text mode, universal newlines): text before the first '>' line is skipped; every
// sequence line is right-stripped and joined; spaces and '\r' are removed; with
// reverse_complement each RECORD is reverse-complemented on its own (IUPAC DNA table, case
// preserved, unknown characters unchanged; 'U' without 'T' -> RNA table; both -> error) and the
// records are concatenated in file order.  snacc_amd/fasta.py is the Python statement of the same
// rules; tests/test_fasta_native.py checks the two against each other.
// Byte-level caveat: right-stripping knows ASCII white space and 0x1c-0x1f, not Unicode spaces.
#include "snacc_hip.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Tables {
    uint8_t dna[256], rna[256];
    Tables()
    {
        for (int i = 0; i < 256; ++i) dna[i] = rna[i] = (uint8_t)i;
        const char *k = "ACGTMRWSYKVHDBXN", *vd = "TGCAKYWSRMBDHVXN";
        for (int i = 0; k[i]; ++i) {
            dna[(uint8_t)k[i]] = (uint8_t)vd[i];
            dna[(uint8_t)(k[i] | 0x20)] = (uint8_t)(vd[i] | 0x20);
        }
        const char *kr = "ACGUMRWSYKVHDBXN", *vr = "UGCAKYWSRMBDHVXN";
        for (int i = 0; kr[i]; ++i) {
            rna[(uint8_t)kr[i]] = (uint8_t)vr[i];
            rna[(uint8_t)(kr[i] | 0x20)] = (uint8_t)(vr[i] | 0x20);
        }
    }
};
const Tables kTables;

inline bool is_space(uint8_t c)
{
    return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f);
}

// returns 0, SNK_E_MIXED
int finish_record(std::vector<uint8_t> &out, size_t start, bool rc)
{
    if (!rc || out.size() == start) return SNK_OK;
    bool has_u = false, has_t = false;
    for (size_t i = start; i < out.size(); ++i) {
        const uint8_t c = out[i];
        has_u |= (c == 'U' || c == 'u');
        has_t |= (c == 'T' || c == 't');
    }
    if (has_u && has_t) return SNK_E_MIXED;
    const uint8_t *t = has_u ? kTables.rna : kTables.dna;
    size_t i = start, j = out.size() - 1;
    while (i < j) {
        const uint8_t a = t[out[i]], b = t[out[j]];
        out[i++] = b; out[j--] = a;
    }
    if (i == j) out[i] = t[out[i]];
    return SNK_OK;
}

int extract(const char *path, bool rc, std::vector<uint8_t> &out, std::string &err)
{
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return SNK_E_ARG; }
    std::vector<uint8_t> buf;
    if (fseek(f, 0, SEEK_END) == 0) {
        long sz = ftell(f);
        if (sz > 0) buf.reserve((size_t)sz);
        fseek(f, 0, SEEK_SET);
    }
    {
        uint8_t tmp[1 << 16];
        size_t r;
        while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
    }
    fclose(f);
    out.clear();
    out.reserve(buf.size());
    const uint8_t *p = buf.data(), *end = p + buf.size();
    bool in_record = false;
    size_t rec_start = 0;
    while (p < end) {
        // one line, universal newlines: \n, \r\n, \r
        const uint8_t *q = p;
        while (q < end && *q != '\n' && *q != '\r') ++q;
        const uint8_t *line_end = q;
        if (q < end) { if (*q == '\r' && q + 1 < end && q[1] == '\n') q += 2; else q += 1; }
        if (p < line_end && *p == '>') {
            if (in_record) {
                int rcode = finish_record(out, rec_start, rc);
                if (rcode) { err = std::string("Mixed RNA/DNA found in ") + path; return rcode; }
            }
            in_record = true;
            rec_start = out.size();
        } else if (in_record) {
            const uint8_t *e = line_end;
            while (e > p && is_space(e[-1])) --e;
            for (const uint8_t *c = p; c < e; ++c)
                if (*c != ' ') out.push_back(*c);
        }
        p = q;
    }
    if (in_record) {
        int rcode = finish_record(out, rec_start, rc);
        if (rcode) { err = std::string("Mixed RNA/DNA found in ") + path; return rcode; }
    }
    if (out.empty()) {
        err = std::string("No sequence extracted. Ensure that file ") + path +
              " contains a proper FASTA definition line (i.e. a line that starts with '>sequence_name').";
        return SNK_E_EMPTY;
    }
    return SNK_OK;
}

thread_local std::string g_fasta_error;

} // namespace


// ---- CSV text of the NCD matrix ---------------------------------------------------------------------
// One field as pandas' to_csv writes a float64 (ref:snacc/cli.py:138-142), i.e. Python's repr: the shortest digit string that
// round-trips (std::to_chars gives the same digits: shortest, and the closest to the value among the shortest), exponent
// form when the decimal exponent is < -4 or >= 16 (at least two exponent digits, always a sign), else positional with
// ".0" for whole numbers; "inf" / "-inf"; NaN is an empty field.  At most 24 bytes.
static size_t snk_repr_f64(double v, char *out)
{
    if (v != v) return 0;
    char b[40];
    char *o = out;
    if (v == 0.0) { if (std::signbit(v)) *o++ = '-'; memcpy(o, "0.0", 3); return (size_t)(o + 3 - out); }
    if (v > 1.7976931348623157e308 || v < -1.7976931348623157e308) { if (v < 0) *o++ = '-'; memcpy(o, "inf", 3); return (size_t)(o + 3 - out); }
    const std::to_chars_result r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);
    // b = [-]d[.ddd]e[+-]XX[X]
    const char *p = b;
    if (*p == '-') { *o++ = '-'; ++p; }
    const char *e = p;
    while (e < r.ptr && *e != 'e') ++e;
    char digits[24]; int nd = 0;
    for (const char *q = p; q < e; ++q) if (*q != '.') digits[nd++] = *q;
    int ex = 0; { const char *q = e + 1; const bool neg = (*q == '-'); ++q; for (; q < r.ptr; ++q) ex = ex * 10 + (*q - '0'); if (neg) ex = -ex; }
    if (ex < -4 || ex >= 16) {                                   // d[.ddd]e[+-]XX
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, (size_t)(nd - 1)); o += nd - 1; }
        *o++ = 'e'; *o++ = ex < 0 ? '-' : '+';
        int ax = ex < 0 ? -ex : ex;
        char t[4]; int nt = 0; do { t[nt++] = (char)('0' + ax % 10); ax /= 10; } while (ax);
        if (nt < 2) t[nt++] = '0';
        while (nt) *o++ = t[--nt];
    } else if (ex < 0) {                                         // 0.000ddd
        *o++ = '0'; *o++ = '.';
        for (int k = 0; k < -ex - 1; ++k) *o++ = '0';
        memcpy(o, digits, (size_t)nd); o += nd;
    } else {                                                     // ddd[.ddd] / ddd000.0
        const int ip = ex + 1;                                   // digits in front of the point
        if (nd <= ip) { memcpy(o, digits, (size_t)nd); o += nd; for (int k = nd; k < ip; ++k) *o++ = '0'; *o++ = '.'; *o++ = '0'; }
        else { memcpy(o, digits, (size_t)ip); o += ip; *o++ = '.'; memcpy(o, digits + ip, (size_t)(nd - ip)); o += nd - ip; }
    }
    return (size_t)(o - out);
}

extern "C" {

const char *snk_fasta_last_error(void) { return g_fasta_error.c_str(); }

void snk_free(void *p) { free(p); }

int snk_fasta_extract(const char *path, int reverse_complement, uint8_t **out, uint64_t *out_len)
{
    if (!path || !out || !out_len) { g_fasta_error = "NULL argument"; return SNK_E_ARG; }
    *out = nullptr; *out_len = 0;
    std::vector<uint8_t> v;
    std::string err;
    int rc = extract(path, reverse_complement != 0, v, err);
    if (rc) { g_fasta_error = err; return rc; }
    uint8_t *m = (uint8_t *)malloc(v.size() ? v.size() : 1);
    if (!m) { g_fasta_error = "out of memory"; return SNK_E_HIP; }
    memcpy(m, v.data(), v.size());
    *out = m; *out_len = v.size();
    return SNK_OK;
}

int snk_fasta_extract_many(int n, const char *const *paths, int reverse_complement, int n_threads,
                           uint8_t **outs, uint64_t *lens)
{
    if (n < 0 || (n && (!paths || !outs || !lens))) { g_fasta_error = "bad arguments"; return SNK_E_ARG; }
    for (int i = 0; i < n; ++i) { outs[i] = nullptr; lens[i] = 0; }
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    std::atomic<int> next(0), first_err(0);
    std::vector<std::string> errs((size_t)n_threads);
    std::vector<int> err_idx((size_t)n_threads, -1);
    auto work = [&](int tid) {
        std::vector<uint8_t> v;
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n || first_err.load()) break;
            std::string err;
            int rc = extract(paths[i], reverse_complement != 0, v, err);
            if (!rc) {
                uint8_t *m = (uint8_t *)malloc(v.size() ? v.size() : 1);
                if (!m) { rc = SNK_E_HIP; err = "out of memory"; }
                else { memcpy(m, v.data(), v.size()); outs[i] = m; lens[i] = v.size(); }
            }
            if (rc) {
                int expected = 0;
                if (first_err.compare_exchange_strong(expected, rc)) { errs[(size_t)tid] = err; err_idx[(size_t)tid] = i; }
                break;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    const int rc = first_err.load();
    if (rc) {
        for (int t = 0; t < n_threads; ++t) if (err_idx[(size_t)t] >= 0) g_fasta_error = errs[(size_t)t];
        for (int i = 0; i < n; ++i) { free(outs[i]); outs[i] = nullptr; lens[i] = 0; }
    }
    return rc;
}

int snk_upload_fasta(snk_ctx *ctx, int n, const char *const *paths, int reverse_complement, int n_threads)
{
    if (!ctx) { g_fasta_error = "ctx is NULL"; return SNK_E_ARG; }
    std::vector<uint8_t *> outs((size_t)(n > 0 ? n : 0), nullptr);
    std::vector<uint64_t> lens((size_t)(n > 0 ? n : 0), 0);
    int rc = snk_fasta_extract_many(n, paths, reverse_complement, n_threads, outs.data(), lens.data());
    if (rc) return rc;                                   // message in snk_fasta_last_error()
    rc = snk_upload(ctx, n, (const uint8_t *const *)outs.data(), lens.data());
    for (auto p : outs) free(p);
    if (rc) g_fasta_error = snk_last_error(ctx);
    return rc;
}

int snk_csv_rows_f64(const double *m, uint64_t rows, uint64_t cols, char *out, uint64_t stride, uint32_t *len, int n_threads)
{
    if ((rows && cols && (!m || !out)) || (rows && !len) || stride < cols * SNK_CSV_FIELD_MAX) return SNK_E_ARG;
    std::atomic<uint64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const uint64_t r = next.fetch_add(1);
            if (r >= rows) return;
            char *o = out + r * stride;
            const double *v = m + r * cols;
            for (uint64_t c = 0; c < cols; ++c) {
                if (c) *o++ = ',';
                o += snk_repr_f64(v[c], o);
            }
            len[r] = (uint32_t)(o - (out + r * stride));
        }
    };
    const int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    std::vector<std::thread> ts;
    for (int t = 1; t < nt; ++t) ts.emplace_back(work);
    work();
    for (auto &t : ts) t.join();
    return SNK_OK;
}

// NCD matrix from integer sizes (ref:snacc/cli.py:131-136 + ref:snacc/pairwise_ncd.py:93-111):
//   out[i][j] = (min(P[i][j], P[j][i]) - min(S[i], S[j])) / max(S[i], S[j]),  S = singles + overhead, P = pairs + overhead,
// int64 arithmetic, then ONE float64 division -- the same two correctly rounded operations as the reference's Python
// (int - int, int / int with both below 2^53) and as snacc_amd.matrix.ncd_matrix, which the CPU tests hold it equal to.
// Tiles of 64 x 64 keep the transposed reads in cache; host threads take tile rows.
int snk_ncd_matrix_u32(const uint32_t *singles, const uint32_t *pairs, uint64_t n, uint32_t overhead, double *out, int n_threads)
{
    if (n && (!singles || !pairs || !out)) return SNK_E_ARG;
    const uint64_t TB = 64;
    const uint64_t tiles = (n + TB - 1) / TB;
    std::atomic<uint64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const uint64_t ti = next.fetch_add(1);
            if (ti >= tiles) return;
            const uint64_t i0 = ti * TB, i1 = std::min(n, i0 + TB);
            for (uint64_t j0 = 0; j0 < n; j0 += TB) {
                const uint64_t j1 = std::min(n, j0 + TB);
                for (uint64_t i = i0; i < i1; ++i) {
                    const int64_t si = (int64_t)singles[i] + overhead;
                    for (uint64_t j = j0; j < j1; ++j) {
                        const int64_t sj = (int64_t)singles[j] + overhead;
                        const int64_t pij = (int64_t)pairs[i * n + j] + overhead, pji = (int64_t)pairs[j * n + i] + overhead;
                        const int64_t num = std::min(pij, pji) - std::min(si, sj);
                        out[i * n + j] = (double)num / (double)std::max(si, sj);
                    }
                }
            }
        }
    };
    const int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    std::vector<std::thread> ts;
    for (int t = 1; t < nt && (uint64_t)t < tiles; ++t) ts.emplace_back(work);
    work();
    for (auto &t : ts) t.join();
    return SNK_OK;
}

} // extern "C"
