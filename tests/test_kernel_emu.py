"""The 2-bit kernel's own source (snacc_amd/csrc/snk_fast.hip.h), compiled for the host and run one
lane at a time (tests/emu/), against the oracle: the parse logic is checked here without a GPU; the
`-m gpu` suite then checks the same code as 64-lane waves on the card."""
import numpy as np
import pytest

import oracle
from emu import fast_sizes


def expect(seqs):
    n = len(seqs)
    s = np.zeros(n, np.uint32)
    p = np.zeros((n, n), np.uint32)
    pure = [len(x) > 0 and set(bytes(x)) <= set(b"ACGT") for x in seqs]
    for i in range(n):
        if pure[i] and len(seqs[i]) > 65536:
            s[i] = oracle.lz4f_size(seqs[i])
        for j in range(n):
            if pure[i] and pure[j] and len(seqs[i]) + len(seqs[j]) > 65536:
                p[i, j] = oracle.lz4f_size_pair(seqs[i], seqs[j])
    return s, p


def check(seqs):
    s, p = fast_sizes(seqs)
    es, ep = expect(seqs)
    assert np.array_equal(s, es), (s, es)
    assert np.array_equal(p, ep), np.argwhere(p != ep)[:8].tolist()


def test_emu_ragged_block_edges():
    lens = [65536, 65537, 131072, 200001, 30000, 35536, 12, 65535 + 65536, 131073, 70000]
    check([oracle.lcg_genome(11 + k, n) for k, n in enumerate(lens)])


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
