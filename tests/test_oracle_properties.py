"""Exact-reuse properties the GPU path relies on, checked on the oracle."""
import ctypes

import numpy as np

from oracle.loader import Stream, lib


def _run(buf, n, upto, s=None):
    L = lib()
    if s is None:
        s = Stream()
        L.snk_oracle_stream_init(ctypes.byref(s))
    assert L.snk_oracle_stream_run(ctypes.byref(s), buf.ctypes.data, n, upto, None) == 0
    return s


def test_prefix_blocks_do_not_depend_on_suffix(oracle_mod):
    """Blocks wholly inside x compress identically in x, x+y and x+z, and leave the same table."""
    o = oracle_mod
    x = o.lcg_genome(5, 300000)
    kx = len(x) // 65536 * 65536
    states = []
    for tail in (np.zeros(0, np.uint8), o.lcg_genome(6, 120000), o.lcg_genome(7, 70000)):
        buf = np.concatenate([x, tail])
        s = _run(buf, len(buf), kx)
        states.append((int(s.out), np.array(s.table)))
    for out, tab in states[1:]:
        assert out == states[0][0]
        assert np.array_equal(tab, states[0][1])


def test_resume_from_snapshot_equals_full_run(oracle_mod):
    o = oracle_mod
    x, y = o.lcg_genome(8, 200000), o.lcg_genome(9, 150000)
    buf = np.concatenate([x, y])
    n = len(buf)
    snap = _run(x, len(x), len(x) // 65536 * 65536)            # computed from x alone
    resumed = Stream()
    ctypes.memmove(ctypes.byref(resumed), ctypes.byref(snap), ctypes.sizeof(Stream))
    _run(buf, n, n, resumed)
    assert int(resumed.out) + 4 == o.lz4f_size(buf)


def test_stats_match_survey_workload(oracle_mod):
    """SURVEY.md 8d workload statistics for 1 Mbp uniform ACGT (sanity of the restatement)."""
    from oracle.loader import lz4f_size_stats
    size, st = lz4f_size_stats(oracle_mod.lcg_genome(1, 1000000))
    assert size == 567799
    assert 180000 < st["sequences"] < 185000
    assert st["too_far"] == 0 and st["bailouts"] == 0
    assert 0.20 < (st["search_probes"] + st["chain_probes"]) / 1e6 < 0.22


def test_u16_bitmap_table_scheme_of_the_2bit_kernel(oracle_mod):
    """The 2-bit kernel keeps, per hash slot, a 16-bit offset inside the current block plus a "written in this block"
    bit, and ages the table at block transitions (tests/lz4_table_model.py).  Run through a plain Python LZ4-frame
    parse, that scheme and liblz4's absolute-position table give the oracle's sizes -- on random genomes, tandem
    repeats with mutations (blocks that end in matches) and a low-complexity run, three blocks and more each."""
    import numpy as np
    from lz4_table_model import U16BitmapTable, PlainTable, frame_size
    rng = np.random.default_rng(2510)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rep = np.tile(rng.choice(acgt, 900), 160)[:140000].copy()
    hit = rng.random(len(rep)) < 0.02
    rep[hit] = rng.choice(acgt, int(hit.sum()))
    seqs = [oracle_mod.lcg_genome(3, 140001), rep, np.repeat(rng.choice(acgt, 5000), 30)[:135000].copy(),
            np.concatenate([oracle_mod.lcg_genome(4, 70000), oracle_mod.lcg_genome(5, 66000)])]
    for a in seqs:
        exp = oracle_mod.lz4f_size(a)
        assert frame_size(a, PlainTable()) == exp
        assert frame_size(a, U16BitmapTable()) == exp
