import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


LCG_A, LCG_C = 6364136223846793005, 1442695040888963407


def lcg_bytes(seed, n, alphabet):
    """Generator used by tests/golden/make_golden.py for non-ACGT inputs (pure Python)."""
    s, out, k = seed, bytearray(n), len(alphabet)
    for i in range(n):
        s = (s * LCG_A + LCG_C) & ((1 << 64) - 1)
        out[i] = alphabet[(s >> 33) % k]
    return bytes(out)


def make_seq(oracle, kind, seed, length):
    """FASTA-set sequence kinds of tests/golden/make_golden.py (CLI_SETS)."""
    if kind == "lcg":
        return bytes(oracle.lcg_genome(seed, length)).decode()
    if kind.startswith("mut:"):
        return bytes(oracle.lcg_mutant(oracle.lcg_genome(int(kind[4:]), length), seed)).decode()
    if kind == "lower":
        return bytes(oracle.lcg_genome(seed, length)).decode().lower()
    if kind == "nrun":
        s = bytearray(bytes(oracle.lcg_genome(seed, length)))
        for start in range(1000, length - 200, 5000):
            s[start:start + 137] = b"N" * 137
        return s.decode()
    raise ValueError(kind)


def write_fasta(path, records, width=80, newline="\n"):
    with open(path, "w", newline="") as f:
        for title, seq in records:
            f.write(">" + title + newline)
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + newline)


def materialise_cli_set(oracle, spec, directory):
    directory.mkdir(parents=True, exist_ok=True)
    for fname, recs in spec["files"].items():
        write_fasta(directory / fname, [(t, make_seq(oracle, k, s, n)) for (t, k, s, n) in recs],
                    newline=spec["newline"])
    return directory
