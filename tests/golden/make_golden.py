#!/usr/bin/env python3
"""Generate tests/golden/golden.json  (run in the build container only; never on the GPU box).

What is recorded (data only -- inputs and expected outputs):

1. liblz4 1.9.3 ``LZ4F_compressFrame(prefs=NULL)`` frame sizes (the binary of this image,
   via ctypes) for generator-defined inputs: the SURVEY.md 8c LCG genomes, ragged lengths
   around the 64 KiB block edge, mutants, non-ACGT and incompressible inputs, and the
   frame-format known answers.
2. Outputs of the REFERENCE's own Python (``/root/reference/snacc/pairwise_ncd.py`` and
   ``cli.py``, imported unmodified) for its fixture ``test_dataset/sample.fa`` and for small
   synthetic FASTA sets: ``extract_sequences``, ``compressed_size`` (all codecs),
   ``compute_distance`` known answers, the CSV matrix the CLI writes with ``-c lz4`` with
   and without ``-r``, and one run of the CLI with a stdlib codec (``-c gzip``: CSV, banners, log).

The reference imports two third-party modules that are absent from this image and cannot be
fetched (``lz4framed``, ``Bio``).  To run its glue code here they are replaced IN MEMORY by
stand-ins defined below: ``lz4framed.compress`` -> liblz4 1.9.3 ``LZ4F_compressFrame`` with
NULL preferences (the dependency's own codec), ``Bio.SeqIO.parse`` -> a minimal
FASTA reader.  Consequently the golden lz4 sizes pin the oracle to liblz4 1.9.3, NOT to a
real py-lz4framed wheel ("parity unpinned" against the wheel: DESIGN.md).  Nothing from the
reference is copied into the repository; only its outputs are stored.
"""
import ctypes
import io
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import liblz4_ref  # noqa: E402  (ctypes binding to the liblz4 binary)

REF = Path("/root/reference")
LCG_A, LCG_C = 6364136223846793005, 1442695040888963407
M64 = (1 << 64) - 1


def lcg_genome(seed, n):
    s, out = seed, bytearray(n)
    for i in range(n):
        s = (s * LCG_A + LCG_C) & M64
        out[i] = b"ACGT"[(s >> 33) & 3]
    return bytes(out)


def lcg_mutant(src, seed):
    s, out = seed, bytearray(src)
    for i in range(len(src)):
        s = (s * LCG_A + LCG_C) & M64
        if (s >> 40) % 50 == 0:
            out[i] = b"ACGT"[(s >> 33) & 3]
    return bytes(out)


def lcg_bytes(seed, n, alphabet):
    s, out = seed, bytearray(n)
    k = len(alphabet)
    for i in range(n):
        s = (s * LCG_A + LCG_C) & M64
        out[i] = alphabet[(s >> 33) % k]
    return bytes(out)


# ---------------------------------------------------------------------------------------------
# in-memory stand-ins for the two absent third-party modules (see module docstring)
# ---------------------------------------------------------------------------------------------
def install_standins():
    lz4framed = types.ModuleType("lz4framed")
    lz4framed.__version__ = "stand-in (liblz4 %s LZ4F_compressFrame, NULL prefs)" % liblz4_ref.version()
    lz4framed.compress = lambda b: liblz4_ref.compress_frame(b)
    sys.modules["lz4framed"] = lz4framed

    class _Seq(str):
        _T = str.maketrans("ACGTMRWSYKVHDBXNacgtmrwsykvhdbxn", "TGCAKYWSRMBDHVXNtgcakywsrmbdhvxn")

        def reverse_complement(self):
            return _Seq(self.translate(self._T)[::-1])

    class _Rec:
        def __init__(self, title, seq):
            self.id, self.seq = title, _Seq(seq)

    def parse(path, fmt):
        assert fmt == "fasta"
        title, lines = None, []
        with open(path, "r") as h:
            for line in h:
                if line.startswith(">"):
                    if title is not None:
                        yield _Rec(title, "".join(lines).replace(" ", "").replace("\r", ""))
                    title, lines = line[1:].rstrip(), []
                elif title is not None:
                    lines.append(line.rstrip())
        if title is not None:
            yield _Rec(title, "".join(lines).replace(" ", "").replace("\r", ""))

    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")
    seqio.parse = parse
    bio.SeqIO = seqio
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = seqio


def write_fasta(path, records, width=80, newline="\n"):
    with open(path, "w", newline="") as f:
        for title, seq in records:
            f.write(">" + title + newline)
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + newline)


# FASTA set specifications: name -> list of (title, kind, seed, length)
#   kind: "lcg" uniform ACGT | "mut:<seed_of_parent>" 2% mutant | "lower" lower-case lcg | "nrun" lcg with N runs
CLI_SETS = {
    "acgt_small": {
        "newline": "\n",
        "files": {
            "g0.fasta": [("g0", "lcg", 1, 30000)],
            "g1.fasta": [("g1", "lcg", 2, 30000)],
            "g2.fna":   [("g2a", "lcg", 3, 20000), ("g2b", "lcg", 4, 15000)],
            "g3.fa":    [("g3", "mut:1", 5, 30000)],
        },
    },
    "ragged_blocks": {
        "newline": "\r\n",
        "files": {
            "a.fasta": [("a", "lcg", 11, 65536)],
            "b.fasta": [("b", "lcg", 12, 70001)],
            "c.fasta": [("c", "lcg", 13, 1000)],
            "d.fsa":   [("d1", "lcg", 14, 40000), ("d2", "lower", 15, 3000), ("d3", "nrun", 16, 30000)],
            "e.fasta": [("e", "mut:12", 17, 70001)],
        },
    },
}


def make_seq(kind, seed, length):
    if kind == "lcg":
        return lcg_genome(seed, length).decode()
    if kind.startswith("mut:"):
        return lcg_mutant(lcg_genome(int(kind[4:]), length), seed).decode()
    if kind == "lower":
        return lcg_genome(seed, length).decode().lower()
    if kind == "nrun":
        s = bytearray(lcg_genome(seed, length))
        for start in range(1000, length - 200, 5000):
            s[start:start + 137] = b"N" * 137
        return s.decode()
    raise ValueError(kind)


def main():
    assert liblz4_ref.available() and liblz4_ref.version() == "1.9.3", "needs the image's liblz4 1.9.3"
    gold = {"liblz4_version": liblz4_ref.version(), "generator": "tests/golden/make_golden.py"}

    # ---- 1. raw frame sizes from liblz4 ---------------------------------------------------------
    fs = liblz4_ref.frame_size
    sizes = {"frames_hex": {"empty": liblz4_ref.compress_frame(b"").hex(),
                            "ACGTx10": liblz4_ref.compress_frame(b"ACGT" * 10).hex()}}
    lcg = []
    for n in (1000, 100000, 1000000):
        x, y = lcg_genome(1, n), lcg_genome(2, n)
        z = lcg_mutant(x, 3)
        lcg.append({"n": n, "x": fs(x), "y": fs(y), "z": fs(z), "xy": fs(x + y), "yx": fs(y + x),
                    "xx": fs(x + x), "xz": fs(x + z), "zx": fs(z + x)})
    sizes["lcg_seed1_2_mut3"] = lcg
    ragged = []
    for n in (0, 1, 4, 12, 13, 14, 100, 65535, 65536, 65537, 65546, 65547, 65548, 131071, 131072, 131073,
              196608, 200001):
        ragged.append({"n": n, "seed": 21, "size": fs(lcg_genome(21, n))})
    sizes["lcg_ragged"] = ragged
    other = []
    for name, alpha, seed, n in (("aa20", b"ACDEFGHIKLMNPQRSTVWY", 31, 150000),
                                 ("bytes256", bytes(range(256)), 32, 150000),
                                 ("acgtn", b"ACGTN", 33, 150000),
                                 ("two", b"AC", 34, 150000)):
        other.append({"name": name, "alphabet_hex": alpha.hex(), "seed": seed, "n": n,
                      "size": fs(lcg_bytes(seed, n, alpha))})
    sizes["other_alphabets"] = other
    mixed = lcg_bytes(41, 100000, bytes(range(256))) + lcg_genome(42, 150000) + lcg_bytes(43, 70000, bytes(range(256)))
    sizes["mixed_raw_then_dna"] = {"parts": [["bytes256", 41, 100000], ["lcg", 42, 150000], ["bytes256", 43, 70000]],
                                   "size": fs(mixed)}
    gold["liblz4_frame_sizes"] = sizes

    # ---- 2. the reference's own Python ------------------------------------------------------------
    install_standins()
    sys.path.insert(0, str(REF))
    import snacc.pairwise_ncd as rp          # the reference, unmodified
    import snacc.cli as rcli
    from click.testing import CliRunner

    kats = [[1174721, 1173133, 1242873, 1242873]]     # ref:docs/NCD Demo.ipynb cells 10/13
    rng = np.random.default_rng(2025)
    for _ in range(40):
        x, y = (int(v) for v in rng.integers(50, 2_000_000, 2))
        cxy, cyx = (int(v) for v in rng.integers(max(x, y), x + y + 100, 2))
        kats.append([x, y, cxy, cyx])
    kats += [[500, 500, 900, 950], [7, 3, 9, 8], [3, 7, 8, 9]]
    gold["compute_distance"] = [{"args": k, "result": rp.compute_distance(*k)} for k in kats]

    sample = REF / "test_dataset" / "sample.fa"
    gold["sample_fa"] = {
        "extract": rp.extract_sequences(sample),
        "extract_rc": rp.extract_sequences(sample, reverse_complement=True),
        "extract_pair": rp.extract_sequences((sample, sample)),
        "sizes_single": {a: rp.compressed_size(sample, a)[1] for a in ("lz4", "gzip", "zlib", "bzip2", "lzma")},
        "sizes_selfpair": {a: rp.compressed_size((sample, sample), a)[1] for a in ("lz4", "gzip", "zlib", "bzip2", "lzma")},
        "sizes_single_rc_lz4": rp.compressed_size(sample, "lz4", reverse_complement=True)[1],
    }

    cli_out = {}
    for set_name, spec in CLI_SETS.items():
        with tempfile.TemporaryDirectory() as td:
            d = Path(td) / "fa"
            d.mkdir()
            for fname, recs in spec["files"].items():
                write_fasta(d / fname, [(t, make_seq(k, s, n)) for (t, k, s, n) in recs], newline=spec["newline"])
            entry = {}
            for rc in (False, True):
                out = Path(td) / ("out_rc.csv" if rc else "out.csv")
                args = [str(d), "-o", str(out), "-c", "lz4", "-n", "1", "--no-show-progress", "--no-log"]
                if rc:
                    args.append("-r")
                cwd = os.getcwd()
                os.chdir(td)
                try:
                    res = CliRunner().invoke(rcli.cli, args)
                finally:
                    os.chdir(cwd)
                assert res.exit_code == 0, res.output
                text = out.read_text().replace(str(d) + "/", "{DIR}/")
                entry["csv_rc" if rc else "csv"] = text
            cli_out[set_name] = entry
    gold["cli_lz4"] = {"sets": {k: {"newline": v["newline"], "files": v["files"]} for k, v in CLI_SETS.items()},
                       "outputs": cli_out}

    # the reference CLI with a stdlib codec (CPU pass-through flow) on the small set, incl. the run log
    with tempfile.TemporaryDirectory() as td:
        d = Path(td) / "fa"
        d.mkdir()
        spec = CLI_SETS["acgt_small"]
        for fname, recs in spec["files"].items():
            write_fasta(d / fname, [(t, make_seq(k, s, n)) for (t, k, s, n) in recs], newline=spec["newline"])
        out = Path(td) / "gz.csv"
        cwd = os.getcwd()
        os.chdir(td)
        try:
            res = CliRunner().invoke(rcli.cli, [str(d), "-o", str(out), "-c", "gzip", "-n", "2", "--no-show-progress"])
        finally:
            os.chdir(cwd)
        assert res.exit_code == 0, res.output
        log_lines = (Path(td) / "gz.md").read_text().splitlines()
        gold["cli_gzip"] = {"set": "acgt_small", "csv": out.read_text().replace(str(d) + "/", "{DIR}/"),
                            "stdout": res.output,
                            # run-log lines that do not depend on time, versions or the output path
                            "log_fixed_lines": [ln.replace(str(d) + "/", "{DIR}/") for ln in log_lines
                                                if not ln.startswith(("* Analysis", "* Output", "* Python", "* snacc", "* py-lz4framed"))]}

    # ---- 3. gzip / zlib sizes from the codec the reference calls (ref:snacc/pairwise_ncd.py:73-78) -------
    import gzip
    import zlib
    dfl = {"zlib_version": zlib.ZLIB_RUNTIME_VERSION, "python": sys.version.split()[0], "cases": []}
    for n in (1000, 100000, 1000000):
        x, y = lcg_genome(1, n), lcg_genome(2, n)
        z = lcg_mutant(x, 3)
        row = {"n": n}
        for name, data in (("x", x), ("y", y), ("z", z), ("xy", x + y), ("yx", y + x), ("xx", x + x), ("xz", x + z)):
            row[name] = {"gzip": len(gzip.compress(data)), "zlib": len(zlib.compress(data))}
        dfl["cases"].append(row)
    dfl["ragged"] = [{"n": n, "seed": 21, "gzip": len(gzip.compress(lcg_genome(21, n))), "zlib": len(zlib.compress(lcg_genome(21, n)))}
                     for n in (0, 1, 2, 3, 4, 100, 32768, 65274, 65275, 65536, 65537, 65798, 98304, 131073, 200001)]
    dfl["other_alphabets"] = [{"name": name, "alphabet_hex": alpha.hex(), "seed": seed, "n": n,
                               "gzip": len(gzip.compress(lcg_bytes(seed, n, alpha))), "zlib": len(zlib.compress(lcg_bytes(seed, n, alpha)))}
                              for name, alpha, seed, n in (("aa20", b"ACDEFGHIKLMNPQRSTVWY", 31, 150000),
                                                          ("bytes256", bytes(range(256)), 32, 150000),
                                                          ("acgtn", b"ACGTN", 33, 150000), ("two", b"AC", 34, 150000))]
    gold["deflate_sizes"] = dfl

    out_path = Path(__file__).with_name("golden.json")
    out_path.write_text(json.dumps(gold, indent=1, sort_keys=True) + "\n")
    print("wrote", out_path, out_path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
