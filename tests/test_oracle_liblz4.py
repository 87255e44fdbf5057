"""Differential fuzz of the C restatement against the liblz4 1.9.3 BINARY of the image
(the third-party codec behind lz4framed.compress, ref:snacc/pairwise_ncd.py:80).
Skipped where the binary is absent; the golden vectors then carry the pin."""
import numpy as np
import pytest

from oracle import liblz4_ref

pytestmark = pytest.mark.skipif(not liblz4_ref.available(), reason="liblz4 binary not present")

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _gen(rng, kind, n):
    if kind == "acgt":
        return rng.choice(ACGT, n)
    if kind == "rand256":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "aa20":
        return rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), n)
    if kind == "alpha64":
        return rng.integers(32, 96, n, dtype=np.uint8)
    if kind == "runs":
        out, tot = [], 0
        while tot < n:
            ln = int(rng.integers(1, 2000))
            out.append(np.full(ln, rng.integers(65, 70), dtype=np.uint8))
            tot += ln
        return np.concatenate(out)[:n]
    if kind == "repeat":
        unit = rng.choice(ACGT, int(rng.integers(1, 5000)))
        a = np.tile(unit, n // len(unit) + 1)[:n].copy()
        m = rng.random(n) < 0.01
        a[m] = rng.choice(ACGT, int(m.sum()))
        return a
    if kind == "mixed":
        parts, tot = [], 0
        while tot < n:
            k = str(rng.choice(["acgt", "rand256", "aa20", "alpha64", "runs", "repeat"]))
            ln = int(rng.integers(1, 150000))
            parts.append(_gen(rng, k, ln))
            tot += ln
        return np.concatenate(parts)[:n]
    raise ValueError(kind)


def test_version():
    assert liblz4_ref.version() == "1.9.3"


def test_edge_lengths(oracle_mod):
    rng = np.random.default_rng(1)
    for n in list(range(0, 40)) + [65535, 65536, 65537, 65546, 65547, 65548, 131071, 131072, 131073,
                                   2 * 65536 + 5, 3 * 65536 + 12, 3 * 65536 + 13]:
        for kind in ("acgt", "rand256", "runs"):
            d = _gen(rng, kind, n) if n else np.zeros(0, dtype=np.uint8)
            assert oracle_mod.lz4f_size(d) == liblz4_ref.frame_size(d), (kind, n)


@pytest.mark.parametrize("kind", ["acgt", "rand256", "aa20", "alpha64", "runs", "repeat", "mixed"])
def test_fuzz(kind, oracle_mod):
    rng = np.random.default_rng(hash(kind) % 2**32)
    for it in range(40):
        c = it % 3
        n = (int(rng.integers(0, 70000)) if c == 0 else
             int(65536 * rng.integers(1, 5) + rng.integers(-20, 20)) if c == 1 else
             int(rng.integers(65536, 500000)))
        d = _gen(rng, kind, n)
        assert oracle_mod.lz4f_size(d) == liblz4_ref.frame_size(d), (kind, n)


def test_raw_block_then_compressible_keeps_partial_table(oracle_mod):
    """liblz4 abandons an incompressible block part-way (limitedOutput) and the hash table keeps
    only the insertions made so far; later linked blocks see that state."""
    from oracle.loader import lz4f_size_stats
    rng = np.random.default_rng(99)
    hits = 0
    for _ in range(25):
        d = _gen(rng, "mixed", int(rng.integers(200000, 500000)))
        r, st = lz4f_size_stats(d)
        assert r == liblz4_ref.frame_size(d)
        hits += st["bailouts"] > 0 and st["sequences"] > 1000
    assert hits >= 5


def test_frame_decoder_against_liblz4_frames(oracle_mod):
    """The tests' own LZ4 frame decoder (used for the GPU round-trip property) decodes liblz4's frames."""
    from lz4_decode import decode_frame
    rng = np.random.default_rng(17)
    for kind, n in (("acgt", 0), ("acgt", 37), ("acgt", 70000), ("acgt", 200001), ("runs", 150000),
                    ("mixed", 300000), ("rand256", 70000), ("repeat", 131072)):
        d = _gen(rng, kind, n) if n else np.zeros(0, dtype=np.uint8)
        assert decode_frame(liblz4_ref.compress_frame(d)) == d.tobytes(), (kind, n)
