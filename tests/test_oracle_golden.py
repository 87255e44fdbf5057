"""The CPU oracle against the committed golden vectors (liblz4 1.9.3 outputs recorded by
tests/golden/make_golden.py) -- this is what pins the oracle on machines without liblz4."""
import numpy as np

from conftest import lcg_bytes


def test_frame_format_kats(golden, oracle_mod):
    fr = golden["liblz4_frame_sizes"]["frames_hex"]
    assert oracle_mod.lz4f_size(b"") == len(bytes.fromhex(fr["empty"])) == 11
    assert oracle_mod.lz4f_size(b"ACGT" * 10) == len(bytes.fromhex(fr["ACGTx10"])) == 29
    # header of the empty frame: magic, FLG=0x60, BD=0x40 (64 KiB), HC, end mark
    assert fr["empty"] == "04224d1860408200000000"


def test_lcg_generator_prefix(oracle_mod):
    # seed 1 starts GCAGGTGGCGTGAAGAGTACTCGGCAAACATG
    assert bytes(oracle_mod.lcg_genome(1, 32)) == b"GCAGGTGGCGTGAAGAGTACTCGGCAAACATG"


def test_lcg_genome_sizes(golden, oracle_mod):
    o = oracle_mod
    for row in golden["liblz4_frame_sizes"]["lcg_seed1_2_mut3"]:
        n = row["n"]
        x, y = o.lcg_genome(1, n), o.lcg_genome(2, n)
        z = o.lcg_mutant(x, 3)
        got = {"x": o.lz4f_size(x), "y": o.lz4f_size(y), "z": o.lz4f_size(z),
               "xy": o.lz4f_size_pair(x, y), "yx": o.lz4f_size_pair(y, x), "xx": o.lz4f_size_pair(x, x),
               "xz": o.lz4f_size_pair(x, z), "zx": o.lz4f_size_pair(z, x)}
        assert got == {k: row[k] for k in got}, n


def test_ragged_lengths(golden, oracle_mod):
    for row in golden["liblz4_frame_sizes"]["lcg_ragged"]:
        assert oracle_mod.lz4f_size(oracle_mod.lcg_genome(row["seed"], row["n"])) == row["size"], row["n"]


def test_other_alphabets_and_raw_blocks(golden, oracle_mod):
    for row in golden["liblz4_frame_sizes"]["other_alphabets"]:
        data = lcg_bytes(row["seed"], row["n"], bytes.fromhex(row["alphabet_hex"]))
        assert oracle_mod.lz4f_size(data) == row["size"], row["name"]
    m = golden["liblz4_frame_sizes"]["mixed_raw_then_dna"]
    parts = []
    for kind, seed, n in m["parts"]:
        parts.append(lcg_bytes(seed, n, bytes(range(256))) if kind == "bytes256"
                     else bytes(oracle_mod.lcg_genome(seed, n)))
    assert oracle_mod.lz4f_size(b"".join(parts)) == m["size"]


def test_ncd_values_of_survey(golden, oracle_mod):
    # n = 1e6, +33: NCD(x,y) = 0.999350356687629, NCD(x,x) = 1.000420899139182
    from snacc_amd.pairwise_ncd import compute_distance
    r = golden["liblz4_frame_sizes"]["lcg_seed1_2_mut3"][2]
    assert r["n"] == 1000000
    x, y, xy, yx, xx = (r[k] + 33 for k in ("x", "y", "xy", "yx", "xx"))
    assert compute_distance(x, y, xy, yx) == 0.999350356687629
    assert compute_distance(x, x, xx, xx) == 1.000420899139182
