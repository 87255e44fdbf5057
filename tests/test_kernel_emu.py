"""The 2-bit kernel's own source (snacc_amd/csrc/snk_fast.hip.h), compiled for the host and run one
lane at a time (tests/emu/) -- and, for the two-lane steady loop every product launch runs, as a lane PAIR on
two host threads in lockstep -- against the oracle: the parse logic is checked here without a GPU; the
`-m gpu` suite then checks the same code as 64-lane waves on the card."""
import numpy as np
import pytest

import oracle
from emu import fast_sizes


def _packable(x, exc_limit):
    """The kernel's admission rule (snk_upload): pure ACGT, or at most 8 + len * exc_limit / 2^20 flagged granules."""
    a = np.frombuffer(bytes(x), dtype=np.uint8)
    if a.size == 0:
        return False
    bad = ~np.isin(a, np.frombuffer(b"ACGT", dtype=np.uint8))
    gran = np.add.reduceat(bad, np.arange(0, a.size, 16)) > 0
    cnt = int(gran.sum())
    return cnt == 0 or (exc_limit > 0 and cnt <= 8 + a.size * exc_limit // 1048576)


def expect(seqs, exc_limit=128):
    n = len(seqs)
    s = np.zeros(n, np.uint32)
    p = np.zeros((n, n), np.uint32)
    pure = [_packable(x, exc_limit) for x in seqs]
    for i in range(n):
        if pure[i] and len(seqs[i]) > 65536:
            s[i] = oracle.lz4f_size(seqs[i])
        for j in range(n):
            if pure[i] and pure[j] and len(seqs[i]) + len(seqs[j]) > 65536:
                p[i, j] = oracle.lz4f_size_pair(seqs[i], seqs[j])
    return s, p


def check(seqs, exc_limit=128):
    s, p = fast_sizes(seqs, exc_limit=exc_limit)
    es, ep = expect(seqs, exc_limit)
    assert np.array_equal(s, es), (s, es)
    assert np.array_equal(p, ep), np.argwhere(p != ep)[:8].tolist()
    # (round 4) the same set through the loop every product launch runs: TWO lanes per chain (snk_fast_steady_spec's C++
    # statement), the emulated lane's partner on a second host thread in lockstep -- with and without exceptions
    s2, p2 = fast_sizes(seqs, exc_limit=exc_limit, spec=True)
    assert np.array_equal(s2, es), ("two lanes", s2, es)
    assert np.array_equal(p2, ep), ("two lanes", np.argwhere(p2 != ep)[:8].tolist())
    if all(_packable(x, 0) or len(x) == 0 for x in seqs):
        # pure ACGT sets: the same pairs again as a FAR chain (table in global memory, u32 absolute positions)
        _, pf = fast_sizes(seqs, exc_limit=exc_limit, far=True)
        assert np.array_equal(pf, ep), ("far", np.argwhere(pf != ep)[:8].tolist())


def test_emu_ragged_block_edges():
    lens = [65536, 65537, 131072, 200001, 30000, 35536, 12, 65535 + 65536, 131073, 70000]
    check([oracle.lcg_genome(11 + k, n) for k, n in enumerate(lens)])


def test_emu_last_blocks_of_every_short_length():
    """(round 4) the loop's own turn-around takes a block step when another full step of the same frame follows: streams of 2 blocks
    plus 0 .. 14 bytes stand on both sides of each of its conditions (lane pair: the wrapper's vote is a pair exchange)."""
    x = oracle.lcg_genome(77, 70000)
    tails = [0, 1, 5, 12, 13, 14]
    check([x] + [oracle.lcg_genome(78 + d, 2 * 65536 - 70000 + d) for d in tails])


def test_emu_lengths_mod_4_and_seams():
    # every residue of len(x) mod 4 (the phase of y inside the packed words) and seams next to block edges
    lens = [100001, 100002, 100003, 100004, 65530, 65533, 131069, 131075, 5, 6, 7]
    check([oracle.lcg_genome(31 + k, n) for k, n in enumerate(lens)])


def test_emu_repeats_long_matches_and_runs():
    rep = [np.tile(oracle.lcg_genome(31, 37), 5000), np.tile(oracle.lcg_genome(32, 5000), 40),
           oracle.lcg_mutant(np.tile(oracle.lcg_genome(32, 5000), 40), 5),
           np.frombuffer(b"A" * 150000, dtype=np.uint8), oracle.lcg_genome(33, 160000),
           np.frombuffer(b"AC" * 40000, dtype=np.uint8)]
    check(rep)


@pytest.mark.parametrize("k0", (0, 2))
def test_emu_long_matches_up_to_and_across_the_seam(k0):
    """(round 4) y starts with a copy of x's last few hundred bases, broken after 13 .. 700 of them: the candidate's run of a long
    match walks through the end of x into y -- 128 bases a step inside one sequence, 16 across the seam -- for two residues of
    len(x) mod 4 per call; the second x ends in a tandem repeat."""
    rng = np.random.default_rng(4128 + k0)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for k in (k0, k0 + 1):
        lx = 66000 + k
        x = oracle.lcg_genome(900 + k, lx).copy()
        if k & 1:
            unit = int(rng.integers(3, 90)); x[-2000:] = np.tile(x[-2000:-2000 + unit], 2000 // unit + 1)[:2000]
        tail = int(rng.integers(300, 1300))
        y = np.concatenate([x[-tail:], oracle.lcg_genome(950 + k, 3000 + 3 * k)])
        cut = int(rng.integers(13, 700))
        y[cut] = acgt[(np.flatnonzero(acgt == y[cut])[0] + 1) % 4]
        seqs += [x, y]
    check(seqs)


def test_emu_relatives():
    a = oracle.lcg_genome(41, 180000)
    check([a, oracle.lcg_mutant(a, 3), oracle.lcg_mutant(a, 4)[1000:], oracle.lcg_genome(42, 90000)])


@pytest.mark.parametrize("seed", range(6))
def test_emu_fuzz(seed):
    rng = np.random.default_rng(1000 + seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for _ in range(int(rng.integers(4, 8))):
        c = int(rng.integers(0, 5))
        n = [int(rng.integers(1, 40)), int(rng.integers(40, 66000)), int(65536 * rng.integers(1, 4) + rng.integers(-20, 21)),
             int(rng.integers(65537, 260000)), int(rng.integers(130000, 140000))][c]
        kind = int(rng.integers(0, 4))
        if kind == 0:
            a = rng.choice(acgt, n)
        elif kind == 1:                                   # tandem repeat with a few substitutions
            unit = rng.choice(acgt, int(rng.integers(1, 3000)))
            a = np.tile(unit, n // len(unit) + 1)[:n].copy()
            m = rng.random(n) < rng.choice([0.0, 0.001, 0.02])
            a[m] = rng.choice(acgt, int(m.sum()))
        elif kind == 2:                                   # runs
            a = np.repeat(rng.choice(acgt, n // 20 + 1), rng.integers(1, 600, n // 20 + 1))[:n].copy()
        else:                                             # low-complexity two-letter stretches inside random
            a = rng.choice(acgt, n)
            for _ in range(3):
                s0 = int(rng.integers(0, max(1, n - 1)))
                ln = len(a[s0:s0 + int(rng.integers(1, 5000))])
                a[s0:s0 + ln] = rng.choice(acgt[:2], ln)
        seqs.append(a)
    if len(seqs[0]) > 200:
        seqs.append(oracle.lcg_mutant(seqs[0], 7)[len(seqs[0]) // 3:])
    check(seqs)


def _with_exceptions(rng, a, runs, singles):
    """N runs and scattered IUPAC codes in an ACGT sequence."""
    a = a.copy()
    n = len(a)
    for _ in range(runs):
        s0 = int(rng.integers(0, n))
        a[s0:s0 + int(rng.integers(1, 700))] = ord("N")
    for _ in range(singles):
        a[int(rng.integers(0, n))] = rng.choice(np.frombuffer(b"NRYKMSWacgtn", dtype=np.uint8))
    return a


def test_emu_exceptions_n_runs_and_iupac():
    """Sequences with a few non-ACGT bytes stay on the 2-bit kernel: the places are served by its byte-accurate
    general path (sentinel entries + overflow table), everything else by the steady loop."""
    rng = np.random.default_rng(17)
    o = oracle
    g = [o.lcg_genome(200 + k, n) for k, n in enumerate([150000, 131072, 90001, 70000, 200003])]
    seqs = [_with_exceptions(rng, g[0], 2, 3), g[1], _with_exceptions(rng, g[2], 1, 0), _with_exceptions(rng, g[3], 0, 5),
            _with_exceptions(rng, g[4], 3, 6), _with_exceptions(rng, o.lcg_mutant(g[0], 9), 1, 2)]
    seqs[2][:40] = ord("N")                          # exceptions at the stream start
    seqs[3][-30:] = ord("N")                         # ... at the very end (seam of every pair with it in front)
    seqs[4][65530:65545] = ord("N")                  # ... across a block edge
    check(seqs, exc_limit=4096)


@pytest.mark.parametrize("seed", range(8))
def test_emu_exceptions_fuzz(seed):
    rng = np.random.default_rng(5000 + seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for _ in range(int(rng.integers(3, 6))):
        n = int(rng.choice([int(rng.integers(66000, 200000)), int(65536 * rng.integers(1, 3) + rng.integers(-20, 21)),
                            int(rng.integers(20000, 66000))]))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            a = rng.choice(acgt, n)
        elif kind == 1:
            unit = rng.choice(acgt, int(rng.integers(1, 2000)))
            a = np.tile(unit, n // len(unit) + 1)[:n].copy()
            m = rng.random(n) < 0.01
            a[m] = rng.choice(acgt, int(m.sum()))
        else:
            a = np.repeat(rng.choice(acgt, n // 20 + 1), rng.integers(1, 300, n // 20 + 1))[:n].copy()
        seqs.append(_with_exceptions(rng, a, int(rng.integers(0, 4)), int(rng.integers(0, 8))))
    seqs.append(_with_exceptions(rng, oracle.lcg_mutant(seqs[0], 7) if set(bytes(seqs[0])) <= set(b"ACGT") else seqs[0].copy(), 1, 1))
    check(seqs, exc_limit=8192)


def _soft_masked(rng, a, pct, run=500):
    """Lower-case stretches of about `run` bases over `pct` % of an ACGT sequence (a soft-masked genome)."""
    a = a.copy()
    n = len(a)
    for s0 in rng.integers(0, max(1, n - run - 200), max(1, n * pct // 100 // run)):
        a[s0:s0 + int(rng.integers(run // 2, run * 3 // 2))] |= 0x20
    return a


def test_emu_soft_masked_stretches():
    """Lower-case stretches are runs of exceptions: the general path walks through them on the real bytes, the table
    entries it leaves are ordinary offsets, and the steady loop compares candidates near them through the mask window."""
    rng = np.random.default_rng(23)
    o = oracle
    g = [o.lcg_genome(300 + k, n) for k, n in enumerate([180000, 140001, 131072, 90000])]
    seqs = [_soft_masked(rng, g[0], 5), _soft_masked(rng, g[1], 20, 300), g[2], _soft_masked(rng, g[3], 2),
            _soft_masked(rng, o.lcg_mutant(g[0], 5), 10)]
    seqs[1][:700] |= 0x20                               # a stretch at the stream start
    seqs[3][-400:] |= 0x20                              # ... and at the end (the seam of every pair with it in front)
    seqs[0][65000:66100] |= 0x20                        # ... across a block edge
    assert all(_packable(x, 65536) for x in seqs)       # (all of them run in the emulated kernel)
    from emu import other_mode_trips
    before = other_mode_trips()
    check(seqs, exc_limit=65536)
    assert other_mode_trips() - before > 20000          # the stretches were walked in the loop's other-case mode


@pytest.mark.parametrize("seed", range(6))
def test_emu_soft_masked_fuzz(seed):
    """The general probe on (code, class) windows: lower-case stretches with n runs, IUPAC codes of both cases and single
    bases of the other case inside and next to them, tandem repeats that cross case boundaries, and relatives in which the
    same region is lower case in one genome and upper case in the other (equal 2-bit codes, different bytes: no match)."""
    rng = np.random.default_rng(9000 + seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = int(rng.choice([int(rng.integers(70000, 160000)), int(65536 * rng.integers(1, 3) + rng.integers(-20, 21))]))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        base = rng.choice(acgt, n)
    elif kind == 1:                                   # tandem repeat with substitutions: long matches across the case boundaries
        unit = rng.choice(acgt, int(rng.integers(3, 900)))
        base = np.tile(unit, n // len(unit) + 1)[:n].copy()
        m = rng.random(n) < 0.004
        base[m] = rng.choice(acgt, int(m.sum()))
    else:
        base = oracle.lcg_genome(9100 + seed, n)
    seqs = [base.copy()]
    for v in range(3):
        a = oracle.lcg_mutant(base, 40 + v)[int(rng.integers(0, 50)):].copy() if v else base.copy()
        a = _soft_masked(rng, a, int(rng.choice([2, 10, 30])), int(rng.choice([60, 300, 900])))
        for _ in range(int(rng.integers(0, 6))):           # n runs / IUPAC codes / other-case singles, in and near the stretches
            s0 = int(rng.integers(0, len(a) - 50))
            what = int(rng.integers(0, 4))
            if what == 0:
                a[s0:s0 + int(rng.integers(1, 40))] = ord("n")
            elif what == 1:
                a[s0] = rng.choice(np.frombuffer(b"RYKMryswN", dtype=np.uint8))
            elif what == 2:
                a[s0] ^= 0x20
            else:
                a[s0:s0 + int(rng.integers(1, 30))] = ord("N")
        seqs.append(a)
    seqs.append(rng.choice(acgt, int(rng.integers(66000, 90000))))
    seqs[-1][2000:2600] |= 0x20
    check(seqs, exc_limit=65536)


def test_emu_lower_case_set():
    """A set in lower case: the same kernel code with the LUTs of the lower-case 5-mers (895 slots: the table's last slot
    stays free) and acgt as the alphabet; upper-case stretches and an n run are its exceptions."""
    o = oracle
    g = [o.lcg_genome(600 + k, n) for k, n in enumerate([150000, 131073, 90000, 200001])]
    seqs = [np.frombuffer(bytes(x).lower(), dtype=np.uint8).copy() for x in g]
    seqs[1][40000:40700] &= 0xDF                        # an upper-case stretch
    seqs[3][70000:70060] = ord("n")
    seqs[3][131000:131200] &= 0xDF
    es = np.array([o.lz4f_size(x) for x in seqs], dtype=np.uint32)
    ep = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    for spec in (False, True):
        s, p = fast_sizes(seqs, exc_limit=65536, lower=True, spec=spec)
        assert np.array_equal(s, es), (spec, s, es)
        assert np.array_equal(p, ep), (spec, np.argwhere(p != ep)[:8].tolist())


def test_emu_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The kernel's source as g++ compiles it for the emulation, built with -fsanitize=address,undefined (no recovery) and run in
    a child interpreter with the ASan runtime preloaded: pure ragged sets, N runs / IUPAC codes, soft-masked stretches -- each
    as one lane and as the lane pair of the two-lane loop.  A sanitizer report ends the child with a non-zero status."""
    import os
    import subprocess
    import sys
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no ASan runtime for this gcc")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "snacc_amd", "csrc")
    so = str(tmp_path / "libfast_emu_san.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-Wno-unused-function", "-DSNK_HOST_EMU", "-I", here, "-I", csrc,
                           "-shared", "-fPIC", "-pthread", "-o", so, os.path.join(here, "fast_emu.cpp")])
    code = f"""
import sys
sys.path.insert(0, {os.path.dirname(here)!r}); sys.path.insert(0, {os.path.dirname(os.path.dirname(here))!r})
import numpy as np
import oracle, emu
import test_kernel_emu as t
emu._SO = {so!r}
emu.build = lambda: emu._SO
rng = np.random.default_rng(77)
t.check([oracle.lcg_genome(11 + k, n) for k, n in enumerate([65536, 65537, 131072, 30000, 12, 65535 + 65536, 100003])])
g = [oracle.lcg_genome(200 + k, n) for k, n in enumerate([150000, 90001, 70000])]
ex = [t._with_exceptions(rng, g[0], 2, 3), g[1], t._with_exceptions(rng, g[2], 1, 4)]
ex[2][:40] = ord("N"); ex[0][65530:65545] = ord("N")
t.check(ex, exc_limit=4096)
sm = [t._soft_masked(rng, g[0], 5), g[1], t._soft_masked(rng, g[2], 20, 300)]
sm[2][-400:] |= 0x20
t.check(sm, exc_limit=65536)
print("sanitized emulation ok")
"""
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "sanitized emulation ok" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
