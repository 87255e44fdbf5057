"""The deflate oracle (oracle/deflate_oracle.c) pinned against the real codec: the
interpreter's own gzip / zlib modules (zlib 1.2.11 in this image), i.e. exactly what ref:snacc/pairwise_ncd.py:73-78
calls.  CPU only."""
import gzip
import zlib

import numpy as np
import pytest

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
SAMPLE = b"ACTGACTAGCTAGCTAACTGGCATCGTAGCTAGCTACGATCATCGATCGTACGTACGTAGATCGATCGATCGTACGATCG"


@pytest.fixture(scope="module")
def D():
    from oracle import deflate
    deflate.lib()
    return deflate


def _gen(rng, kind, n):
    if kind == 0:
        return rng.choice(ACGT, n)
    if kind == 1:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 2:
        u = rng.choice(ACGT, int(rng.integers(1, 2000)))
        a = np.tile(u, n // len(u) + 1)[:n].copy()
        m = rng.random(n) < 0.01
        a[m] = rng.choice(ACGT, int(m.sum()))
        return a
    if kind == 3:
        return rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), n)
    if kind == 4:
        return np.repeat(rng.choice(ACGT, n // 50 + 1), rng.integers(1, 100, n // 50 + 1))[:n].copy()
    return rng.choice(np.frombuffer(b"AC", dtype=np.uint8), n)


def test_reference_fixture_sizes(D):
    # sizes of ref:test_dataset/sample.fa incl. sys.getsizeof's 33 (single / self pair)
    assert D.gzip_size(SAMPLE) + 33 == 89 and D.gzip_size(SAMPLE, SAMPLE) + 33 == 92
    assert D.zlib_size(SAMPLE) + 33 == 77 and D.zlib_size(SAMPLE, SAMPLE) + 33 == 80


def test_empty_and_tiny(D):
    for b in (b"", b"A", b"AC", b"ACG", b"ACGT", b"AAAAAAAAAA", bytes(range(256))):
        assert D.gzip_size(b) == len(gzip.compress(b))
        assert D.zlib_size(b) == len(zlib.compress(b))


def test_differential_fuzz_against_the_zlib_binary(D):
    rng = np.random.default_rng(20261003)
    for it in range(90):
        n = int(rng.integers(0, 150000)) if it % 3 else int(rng.integers(0, 300))
        a = _gen(rng, it % 6, n)
        b = bytes(a)
        assert D.gzip_size(a) == len(gzip.compress(b)), (it, n)
        assert D.zlib_size(a) == len(zlib.compress(b)), (it, n)


def test_lengths_around_window_and_block_edges(D):
    rng = np.random.default_rng(3)
    for n in (32767, 32768, 32769, 65535, 65536, 65537, 65274, 65275, 65536 + 261, 65536 + 262, 98304, 98305, 131072):
        for kind in (0, 2):
            a = _gen(rng, kind, n)
            assert D.gzip_size(a) == len(gzip.compress(bytes(a))), (n, kind)
            assert D.zlib_size(a) == len(zlib.compress(bytes(a))), (n, kind)


def test_pairs_are_concatenations(D, oracle_mod):
    x, y = oracle_mod.lcg_genome(1, 120000), oracle_mod.lcg_genome(2, 90000)
    b = bytes(x) + bytes(y)
    assert D.gzip_size(x, y) == len(gzip.compress(b))
    assert D.zlib_size(x, y) == len(zlib.compress(b))
    assert D.gzip_size(x, np.zeros(0, np.uint8)) == D.gzip_size(x)


def test_block_cost_function_matches_stream(D, oracle_mod):
    # one block: its bits from dfl_oracle_block_bits (histogram only) == the traced block
    x = oracle_mod.lcg_genome(5, 60000)
    raw, sym, blk = D.trace(x, level=9)
    first = sym[:16383]
    lf = np.zeros(286, np.uint16); df = np.zeros(30, np.uint16)
    for s in first:
        s = int(s)
        if s >> 31:
            lc, dist = (s >> 16) & 0x7fff, (s & 0xffff) - 1
            lcode = lc if lc < 8 else (28 if lc == 255 else 4 * (lc.bit_length() - 1) - 4 + ((lc >> (lc.bit_length() - 3)) & 3))
            dcode = dist if dist < 4 else 2 * (dist.bit_length() - 1) + ((dist >> (dist.bit_length() - 2)) & 1)
            lf[257 + lcode] += 1; df[dcode] += 1
        else:
            lf[s] += 1
    assert D.block_bits(lf, df) == int(blk[0])
    assert (int(blk.sum()) + 7) // 8 == raw


def test_golden_deflate_sizes(D, golden, oracle_mod):
    """Committed sizes produced by zlib 1.2.11 (tests/golden/make_golden.py): pins the oracle to that version
    even where the interpreter running the tests links another zlib."""
    g = golden["deflate_sizes"]
    assert g["zlib_version"] == "1.2.11"
    from conftest import lcg_bytes
    for row in g["cases"]:
        n = row["n"]
        x, y = oracle_mod.lcg_genome(1, n), oracle_mod.lcg_genome(2, n)
        z = oracle_mod.lcg_mutant(x, 3)
        parts = {"x": (x, None), "y": (y, None), "z": (z, None), "xy": (x, y), "yx": (y, x), "xx": (x, x), "xz": (x, z)}
        for name, (a, b) in parts.items():
            if n == 1000000 and name not in ("x", "xy", "xz"):
                continue                                   # keep the CPU suite short
            assert D.gzip_size(a, b) == row[name]["gzip"], (n, name)
            assert D.zlib_size(a, b) == row[name]["zlib"], (n, name)
    for r in g["ragged"]:
        a = oracle_mod.lcg_genome(r["seed"], r["n"])
        assert (D.gzip_size(a), D.zlib_size(a)) == (r["gzip"], r["zlib"]), r["n"]
    for r in g["other_alphabets"]:
        a = lcg_bytes(r["seed"], r["n"], bytes.fromhex(r["alphabet_hex"]))
        assert (D.gzip_size(a), D.zlib_size(a)) == (r["gzip"], r["zlib"]), r["name"]


def _stream(D, x, y=None, level=9):
    raw, sym, blk = D.trace(x, y, level)
    ln = np.where(sym >> 31, ((sym >> 16) & 0x7fff) + 3, 1).astype(np.int64)
    pos = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.int64)
    return sym, pos


@pytest.mark.parametrize("level", [9, 6])
def test_symbol_stream_of_a_pair_is_x_prefix_seam_y_suffix(D, oracle_mod, level):
    """The exactness argument of the GPU path, checked on the oracle's own symbol streams:
    the stream of x+y equals x's stream until shortly before the seam, and y's stream from the first position
    >= 32 507 bytes behind the seam at which both parsers stand right behind a match ending at the same place."""
    rng = np.random.default_rng(level)
    x = oracle_mod.lcg_genome(11, 150000)
    rel = x[20000:].copy()
    hit = rng.random(len(rel)) < 0.02
    rel[hit] = rng.choice(ACGT, int(hit.sum()))
    for y in (oracle_mod.lcg_genome(12, 120000), rel, _gen(rng, 3, 90000)):
        lx = len(x)
        sx, px = _stream(D, x, None, level)
        sy, py = _stream(D, y, None, level)
        sp, pp = _stream(D, x, y, level)
        # x's own stream is a prefix of the pair's, at least up to 600 bytes before the seam (the restart point)
        k = int(np.searchsorted(px, lx - 600))
        assert np.array_equal(sx[:k], sp[:k]) and np.array_equal(px[:k], pp[:k])
        # first common "right behind a match" position at least 32 507 bytes behind the seam
        pya = py + lx
        found = None
        for kp in np.flatnonzero(pp >= lx + 32507):
            if sp[kp - 1] >> 31:
                ky = int(np.searchsorted(pya, pp[kp]))
                if ky < len(pya) and pya[ky] == pp[kp] and ky > 0 and (sy[ky - 1] >> 31):
                    found = (int(kp), ky)
                    break
        assert found is not None and pp[found[0]] < lx + 34000      # on these inputs they meet within a few hundred bytes
        kp, ky = found
        assert np.array_equal(sp[kp:], sy[ky:]) and np.array_equal(pp[kp:], pya[ky:])


@pytest.mark.parametrize("level", [9, 6])
def test_a_parser_started_mid_stream_joins_the_true_stream(D, oracle_mod, level):
    """The argument behind the parallel per-sequence pass: a parser that starts at position p0 "as if right behind a
    match" (all earlier positions in the hash chains) produces, from the first position at which it and the real
    parser both stand right behind a match, exactly the real stream."""
    x = oracle_mod.lcg_genome(31, 140000)
    st, pt = _stream(D, x, None, level)
    for p0 in (32768, 65536, 98304, 70001):
        ss = D.trace_from(x, p0, level)
        ln = np.where(ss >> 31, ((ss >> 16) & 0x7fff) + 3, 1).astype(np.int64)
        ps = p0 + np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.int64)
        found = None
        for b in range(len(ss)):
            if b == 0 or (ss[b - 1] >> 31):
                a = int(np.searchsorted(pt, ps[b]))
                if a < len(pt) and pt[a] == ps[b] and a > 0 and (st[a - 1] >> 31):
                    found = (a, b)
                    break
        assert found is not None and ps[found[1]] < p0 + 2048        # well inside the 2 KiB overlap of the segments
        a, b = found
        assert np.array_equal(st[a:], ss[b:]) and np.array_equal(pt[a:], ps[b:])


def _window_base(p0, n):
    """Python statement of dfl_window_base (snacc_amd/csrc/snk_deflate.hip.h): where zlib's window starts when the
    parser stands at loop top p0 of an n-byte stream."""
    base = ((p0 - 65275) >> 15) << 15 if p0 >= 65275 else 0
    while True:
        t = base + (65274 if n <= base + 65535 else 65275)
        if p0 < t:
            return base
        base += 32768


def test_window_slide_positions_closed_form(D, oracle_mod):
    """The GPU path decides "may this block be stored" from a closed form of zlib's window position; the oracle
    slides a real window.  Same position at every block flush, for lengths around every kind of edge."""
    rng = np.random.default_rng(9)
    lens = [1000, 65274, 65275, 65536, 65537, 65798, 98041, 98042, 98043, 98304, 130809, 130810, 131072, 163840, 200001]
    lens += [int(v) for v in rng.integers(60000, 400000, 12)]
    for n in lens:
        for kind in (0, 1):                      # DNA (long blocks) and random bytes (short stored blocks)
            a = _gen(rng, kind, n)
            for p0, base in D.window_trace(a, 9):
                assert _window_base(p0, n) == base, (n, kind, p0, base)


def test_bytes_behind_the_end_of_the_input(D, oracle_mod):
    """zlib's match compare may run past the last input byte into whatever its window still holds.  The GPU path
    (dfl_byte in snk_deflate.hip.h) says: zeros while the window has not slid, else the bytes 32 KiB earlier --
    and the window slides for the first time at the first loop top >= 65 274 that has fewer than 262 bytes left, so a
    stream of 65 275 .. 65 536 bytes ends with a slid window although it fitted the buffer.  Checked against the
    oracle's real window as it stands at the end."""
    rng = np.random.default_rng(17)
    for n in [5, 300, 40000, 65273, 65274, 65275, 65276, 65535, 65536, 65537, 65600, 98000, 98304, 98305, 131071, 131072,
              131073, 200001] + [int(v) for v in rng.integers(65537, 500000, 10)]:
        a = _gen(rng, 1, n)                       # random bytes: any wrong source position shows
        got = D.bytes_behind_end(a, 9)
        if _window_base(n, n) == 0:               # the state at the last loop top (p = n): slid for every n >= 65 274
            exp = np.zeros(258, dtype=np.uint8)
        else:
            exp = a[n - 32768:n - 32768 + 258]
        assert np.array_equal(got[:len(exp)], exp), n


def test_rules_model_of_the_gpu_kernel_matches_the_codec(D, oracle_mod):
    """oracle/deflate_rules.c states the rules the GPU kernel applies -- data-only hash chains, the distance / budget /
    nice-length rules of a chain walk, bytes behind the end, the closed-form window position -- without any of zlib's
    window machinery.  Same sizes as the codec: fuzz sets (all alphabets), window-edge lengths, tails that repeat."""
    from fuzzgen import make_set
    rng = np.random.default_rng(5)
    seqs = [a for seed in range(300, 312) for a in make_set(seed)]
    seqs += [_gen(rng, k, n) for n in (0, 1, 2, 3, 4, 65273, 65274, 65275, 65276, 65535, 65536, 65537, 98304) for k in (0, 1)]
    for n in (65275, 65400, 65536, 70000):                       # a tail that occurs twice before (see the GPU test)
        a = rng.choice(ACGT, n)
        tail, rem = a[n - 40:].copy(), a[n - 32768:n - 32768 + 200].copy()
        a[n - 20040:n - 20000] = tail
        a[n - 20000:n - 19800] = rem
        a[n - 5040:n - 5000] = tail
        seqs.append(a)
    x, y = oracle_mod.lcg_genome(3, 40000), oracle_mod.lcg_genome(4, 25400)
    seqs.append(np.concatenate([x, y]))                          # a pair is a concatenation
    for a in seqs:
        b = bytes(a)
        assert D.rules_raw_size(a, 9) + 18 == len(gzip.compress(b)), len(a)
        assert D.rules_raw_size(a, 6) + 6 == len(zlib.compress(b)), len(a)


def test_six_byte_shortcut_decides_like_the_full_chain_walk(D, oracle_mod):
    """The gzip kernel first looks only at the chain members that share the probe's hash of six bytes and takes their
    result if it is a match of >= 6 bytes.  The rules program replays every probe both ways."""
    from fuzzgen import make_set
    rng = np.random.default_rng(6)
    seqs = [a for seed in range(500, 508) for a in make_set(seed)]
    seqs += [oracle_mod.lcg_genome(9, 300000), rng.choice(np.frombuffer(b"AC", dtype=np.uint8), 120000),
             np.repeat(rng.choice(ACGT, 3000), 40)[:100000].copy(), np.tile(oracle_mod.lcg_genome(8, 700), 200)]
    decided = 0
    for a in seqs:
        for level in (9, 6):
            v, k = D.rules_check_kpass(a, level)
            assert v == 0, (len(a), level, v)
            decided += k
    assert decided > 300000


def test_pair_job_of_the_gpu_path_as_a_cpu_program(D, oracle_mod):
    """oracle/deflate_rules.c, dfl_rules_pair_size: the pair job exactly as the kernel does it -- chains per sequence
    plus the two seam positions, restart from x's own stream, resynchronisation with y's own stream, pricing of the
    spliced symbol stream -- against the codec on all ordered pairs of fuzz sets and on short / empty / edge inputs."""
    from fuzzgen import make_set
    sets = [make_set(602), make_set(20016)]
    sets.append([np.zeros(0, np.uint8), np.frombuffer(b"A", dtype=np.uint8), np.frombuffer(b"ACGTAC", dtype=np.uint8),
                 oracle_mod.lcg_genome(1, 599), oracle_mod.lcg_genome(2, 601), oracle_mod.lcg_genome(3, 33000),
                 oracle_mod.lcg_genome(4, 65536), oracle_mod.lcg_genome(5, 70000)])
    for seqs in sets:
        raw = [bytes(s) for s in seqs]
        for i, a in enumerate(seqs):
            for j, b in enumerate(seqs):
                assert D.rules_pair_size(a, b, 9) + 18 == len(gzip.compress(raw[i] + raw[j])), (len(a), len(b))
                assert D.rules_pair_size(a, b, 6) + 6 == len(zlib.compress(raw[i] + raw[j])), (len(a), len(b))
