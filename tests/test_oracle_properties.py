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
