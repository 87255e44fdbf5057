"""Host side of the drop-in (no GPU): FASTA ingest, reverse complement, the NCD formula, the
matrix assembly and the CSV writer -- against the outputs of the reference's own Python
recorded in tests/golden/golden.json."""
import itertools
from pathlib import Path

import numpy as np
import pytest

from conftest import materialise_cli_set
from snacc_amd import compressed_size, compute_distance
from snacc_amd import fasta
from snacc_amd.cli import discover_files, write_matrix_csv, write_matrix_csv_pandas, write_matrix_csv_python
from snacc_amd.matrix import GETSIZEOF_OVERHEAD, ncd_matrix
from snacc_amd.pairwise_ncd import extract_sequences

SAMPLE_TEXT = (">derice\r\nACTGACTAGCTAGCTAACTG\r\n>sanka\r\nGCATCGTAGCTAGCTACGAT\r\n"
               ">junior\r\nCATCGATCGTACGTACGTAG\r\n>yul\r\nATCGATCGATCGTACGATCG")   # ref:test_dataset/sample.fa


@pytest.fixture()
def sample_fa(tmp_path):
    p = tmp_path / "sample.fa"
    with open(p, "w", newline="") as f:
        f.write(SAMPLE_TEXT)
    return p


def test_extract_sequences_sample(golden, sample_fa):
    g = golden["sample_fa"]
    assert extract_sequences(sample_fa) == g["extract"]
    assert extract_sequences(sample_fa, reverse_complement=True) == g["extract_rc"]
    assert extract_sequences((sample_fa, sample_fa)) == g["extract_pair"]


def test_getsizeof_overhead():
    assert GETSIZEOF_OVERHEAD == 33


def test_stdlib_codecs_match_reference(golden, sample_fa, monkeypatch):
    monkeypatch.setenv("SNACC_DEFLATE", "stdlib")       # gzip / zlib: the reference's own call, asked for explicitly
    g = golden["sample_fa"]
    for algo in ("gzip", "zlib", "bzip2", "lzma"):
        key, size = compressed_size(sample_fa, algo)
        assert key == sample_fa and size == g["sizes_single"][algo]
        key, size = compressed_size((sample_fa, sample_fa), algo)
        assert key == (sample_fa, sample_fa) and size == g["sizes_selfpair"][algo]


def test_unknown_algorithm_is_keyerror(sample_fa):
    with pytest.raises(KeyError):
        compressed_size(sample_fa, "snappy")


def test_empty_fasta_raises_valueerror(tmp_path):
    p = tmp_path / "empty.fa"
    p.write_text("no header here\nACGT\n")
    with pytest.raises(ValueError, match="No sequence extracted"):
        extract_sequences(p)


def test_save_directory_stdlib(tmp_path, sample_fa):
    out = tmp_path / "blobs"
    out.mkdir()
    compressed_size(sample_fa, "gzip", save_directory=out)
    compressed_size((sample_fa, sample_fa), "zlib", save_directory=out)
    assert (out / "sample.fa.gz").exists() and (out / "samplesample.fa.ZLIB").exists()


def test_fasta_reader_details(tmp_path):
    p = tmp_path / "x.fna"
    p.write_text("; comment\n\n>r1 desc\nAC GT\nacgtn \n\n>r2\n>r3\nNNRY\n")
    recs = list(fasta.read_fasta_records(p))
    assert recs == [("r1 desc", "ACGTacgtn"), ("r2", ""), ("r3", "NNRY")]
    assert fasta.read_sequence(p) == "ACGTacgtnNNRY"
    # per-record reverse complement, records kept in order (ref:snacc/pairwise_ncd.py:32-36)
    assert fasta.read_sequence(p, True) == "nacgtACGT" + "" + "RYNN"


def test_reverse_complement_tables():
    assert fasta.reverse_complement("ACGTMRWSYKVHDBXN") == "NXVHDBMRSWYKACGT"
    assert fasta.reverse_complement("acgu") == "acgu"
    assert fasta.reverse_complement("AC-GT*") == "*AC-GT"
    with pytest.raises(ValueError):
        fasta.reverse_complement("ACGTU")


def test_compute_distance_kats(golden):
    for kat in golden["compute_distance"]:
        assert compute_distance(*kat["args"]) == kat["result"], kat["args"]
    assert compute_distance(1174721, 1173133, 1242873, 1242873) == 0.05936728806244206


def test_sequence_level_ncd_wrapper_stdlib_codecs(monkeypatch):
    """``ncd(seq_i, seq_j, compressor)`` (north_star's convenience; not in the reference): the four sizes as
    ref:snacc/pairwise_ncd.py:69-90 obtains them, then the reference's formula -- lzma / bzip2 here; lz4 on the GPU box."""
    import bz2
    import lzma
    import sys as _sys
    import snacc
    from snacc_amd import compute_distance, ncd
    assert snacc.ncd is ncd
    x, y = "ACGTTGCAAGGCTA" * 300, "ACGTTGCTAGGCTAACG" * 250
    for name, fn in (("lzma", lzma.compress), ("bzip2", bz2.compress)):
        c = lambda s: _sys.getsizeof(fn(s.encode()))                     # noqa: E731
        assert ncd(x, y, name) == compute_distance(c(x), c(y), c(x + y), c(y + x))
        assert ncd(x.encode(), y.encode(), name) == ncd(x, y, name)
    monkeypatch.setenv("SNACC_DEFLATE", "stdlib")                           # the reference's own gzip call, asked for explicitly
    import gzip
    c = lambda s: _sys.getsizeof(gzip.compress(s.encode()))              # noqa: E731
    assert ncd(x, y, "gzip") == compute_distance(c(x), c(y), c(x + y), c(y + x))
    with pytest.raises(KeyError):
        ncd(x, y, "zstd")


def test_ncd_matrix_equals_scalar_formula():
    rng = np.random.default_rng(3)
    n = 17
    s = rng.integers(500, 600000, n)
    s[3] = s[5]                                   # exercise the x == y branch off the diagonal
    p = rng.integers(600000, 1300000, (n, n))
    m = ncd_matrix(s, p)
    for i, j in itertools.product(range(n), repeat=2):
        assert m[i, j] == compute_distance(int(s[i]), int(s[j]), int(p[i, j]), int(p[j, i]))
    assert np.array_equal(m, m.T)


def test_native_ncd_assembly_equals_the_numpy_statement_and_the_scalar_formula():
    """snk_ncd_matrix_u32 (the library's host threads; what cli.gpu_matrix and bench.py's matrix run call on the raw uint32
    sizes) is bit-equal to matrix.ncd_matrix -- and through it to the reference's compute_distance -- over tile edges (63 / 64 /
    65 / 129 sequences), equal sizes off the diagonal, the smallest and largest sizes the kernels return, and both overheads
    (33 = getsizeof alone; gzip's sizes carry their wrapper bytes already)."""
    from snacc_amd import hip_backend
    from snacc_amd.matrix import ncd_matrix_raw
    assert hip_backend.load() is not None, "libsnacc_hip.so must load (host code: no GPU needed)"
    rng = np.random.default_rng(21)
    for n in (1, 2, 17, 63, 64, 65, 129, 300):
        s = rng.integers(11, 600000, n).astype(np.uint32)
        p = rng.integers(11, 1300000, (n, n)).astype(np.uint32)
        if n > 5:
            s[3] = s[5]
            p[0, 1] = 0xFFFFFFF0
            s[2] = 0xFFFFFF00
        for overhead in (33, 0):
            if overhead == 0 and n == 1:
                continue
            got = ncd_matrix_raw(s, p, overhead)
            want = ncd_matrix(s.astype(np.int64) + overhead, p.astype(np.int64) + overhead)
            assert got.dtype == np.float64 and got.tobytes() == want.tobytes(), (n, overhead)
        if n == 17:
            m = ncd_matrix_raw(s, p)
            for i, j in itertools.product(range(n), repeat=2):
                assert m[i, j] == compute_distance(int(s[i]) + 33, int(s[j]) + 33, int(p[i, j]) + 33, int(p[j, i]) + 33)
    # other dtypes / shapes take the numpy statement (same numbers)
    s64 = np.array([100, 200], dtype=np.int64)
    p64 = np.array([[150, 260], [250, 330]], dtype=np.int64)
    assert np.array_equal(ncd_matrix_raw(s64, p64, 0), ncd_matrix(s64, p64))
    assert ncd_matrix_raw(np.zeros(0, np.uint32), np.zeros((0, 0), np.uint32)).shape == (0, 0)


def test_csv_writer_formats_in_row_chunks(tmp_path, monkeypatch):
    """The float fields are formatted CSV_CHUNK_ROWS rows at a time into one reused buffer (never the text of the whole
    matrix at once); chunk edges do not show in the file, and a failed allocation falls back to the Python statement."""
    import snacc_amd.cli as cli_mod
    rng = np.random.default_rng(9)
    n = 23
    m = 0.9 + 0.2 * rng.random((n, n))
    files = [Path(f"/d/g{i:03d}.fa") for i in range(n)]
    ref = tmp_path / "ref.csv"
    write_matrix_csv_python(files, m, ref)
    for chunk in (1, 5, 23, 256):
        monkeypatch.setattr(cli_mod, "CSV_CHUNK_ROWS", chunk)
        out = tmp_path / f"c{chunk}.csv"
        write_matrix_csv(files, m, out)
        assert out.read_bytes() == ref.read_bytes(), chunk
    real = cli_mod._csv_row_formatter

    def failing():
        def fmt(rows):
            raise MemoryError
        return fmt if real() is not None else None
    monkeypatch.setattr(cli_mod, "_csv_row_formatter", failing)
    out = tmp_path / "fallback.csv"
    write_matrix_csv(files, m, out)
    assert out.read_bytes() == ref.read_bytes()


@pytest.mark.parametrize("set_name", ["acgt_small", "ragged_blocks"])
@pytest.mark.parametrize("rc", [False, True])
def test_csv_from_oracle_sizes_matches_reference_cli(golden, oracle_mod, tmp_path, set_name, rc):
    """discover -> extract (once per file) -> sizes (oracle as the checker) -> NCD -> CSV must equal,
    byte for byte, what the reference CLI wrote for the same FASTA set."""
    spec = golden["cli_lz4"]["sets"][set_name]
    d = materialise_cli_set(oracle_mod, spec, tmp_path / "fa")
    files = discover_files([str(d)])
    assert [f.name for f in files] == sorted(spec["files"])
    seqs = [extract_sequences(f, reverse_complement=rc).encode() for f in files]
    singles = np.array([oracle_mod.lz4f_size(s) for s in seqs]) + GETSIZEOF_OVERHEAD
    pairs = np.array([[oracle_mod.lz4f_size_pair(a, b) for b in seqs] for a in seqs]) + GETSIZEOF_OVERHEAD
    out = tmp_path / "out.csv"
    write_matrix_csv(files, ncd_matrix(singles, pairs), out)
    want = golden["cli_lz4"]["outputs"][set_name]["csv_rc" if rc else "csv"].replace("{DIR}", str(d))
    assert out.read_text() == want


def test_discover_files_rules(tmp_path):
    d = tmp_path / "in"
    (d / "sub").mkdir(parents=True)
    for name in ("b.FASTA", "a.fa", "c.txt", "d.faa", "sub/e.fa"):
        (d / name).write_text(">x\nACGT\n")
    extra = tmp_path / "z.txt"
    extra.write_text(">x\nACGT\n")
    files = discover_files([str(d), str(extra), str(d / "a.fa")])
    assert [f.name for f in files] == ["a.fa", "b.FASTA", "d.faa", "z.txt"]     # non-recursive, dedup, sorted


def test_direct_csv_writer_equals_pandas(tmp_path):
    """the direct writer must produce the bytes pandas' to_csv produces."""
    rng = np.random.default_rng(5)
    n = 40
    m = rng.random((n, n)) * 1.2
    m[0, 1] = 1.0; m[1, 0] = 1e-7; m[2, 2] = 123456789.125; m[3, 4] = 0.1 + 0.2; m[5, 5] = 1e22; m[6, 7] = 5e-324
    files = [Path(f"/data/set, one/g{i:03d}.fa") if i % 9 == 0 else Path(f"/data/x/g{i:03d}.fasta") for i in range(n)]
    files[3] = Path('/data/we"ird/q.fa')
    files[11] = Path("/data/x/a.b")           # Path order != string order against /data/x/a/...
    files[12] = Path("/data/x/a/b")
    m[8, 8] = float("nan"); m[9, 1] = 0.0; m[9, 2] = -0.0; m[9, 3] = 1e16; m[9, 4] = 9999999999999998.0; m[9, 5] = 1e-4
    m[9, 6] = 9.999e-5; m[9, 7] = 1.7976931348623157e308; m[9, 8] = 2.2250738585072014e-308; m[9, 9] = 100.0; m[9, 10] = 0.5
    a, b, c = tmp_path / "direct.csv", tmp_path / "pandas.csv", tmp_path / "python.csv"
    write_matrix_csv(files, m, a)               # the library's formatter (snk_csv_rows_f64) when the library loads
    write_matrix_csv_pandas(files, m, b)
    write_matrix_csv_python(files, m, c)        # the Python statement (csv module + repr)
    assert a.read_bytes() == b.read_bytes()
    assert c.read_bytes() == b.read_bytes()


def test_native_float_fields_equal_python_repr():
    """snk_csv_rows_f64 writes every float64 as Python's repr does (what pandas' to_csv writes, ref:snacc/cli.py:138-142):
    random doubles over the whole exponent range, the NCD range, powers of ten around both notation switches, subnormals,
    whole numbers, infinities; NaN is an empty field."""
    import struct
    from snacc_amd.cli import _csv_row_bodies
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 2 ** 63, 200000, dtype=np.int64).astype(np.uint64) | (rng.integers(0, 2, 200000).astype(np.uint64) << np.uint64(63))
    vals = np.frombuffer(bits.tobytes(), dtype=np.float64).copy()
    vals = vals[np.isfinite(vals)]
    extra = [10.0 ** k for k in range(-30, 31)] + [1.5 * 10.0 ** k for k in range(-8, 20)] + [float(k) for k in range(-5, 40)]
    extra += [0.1 + 0.2, 1 / 3, 2 / 3, 5e-324, 2.5e-323, float("inf"), float("-inf"), 123456789012345680.0, 1e15 + 0.5, 0.00010000000000000002]
    ncd = 0.9 + 0.2 * rng.random(100000)
    allv = np.concatenate([vals, np.array(extra), ncd, -ncd[:1000]])
    cols = 1000
    n = (len(allv) // cols) * cols
    m = np.ascontiguousarray(allv[:n].reshape(-1, cols))
    m[0, 3] = float("nan")
    bodies = _csv_row_bodies(m)
    assert bodies is not None, "libsnacc_hip.so must load (host code: no GPU needed)"
    for r, body in enumerate(bodies):
        exp = ",".join("" if v != v else repr(v) for v in m[r].tolist())
        assert body.decode() == exp, (r, [(a, b) for a, b in zip(body.decode().split(","), exp.split(",")) if a != b][:3])


def test_cli_gzip_flow_matches_reference_cli(golden, oracle_mod, tmp_path, monkeypatch):
    """`snacc <dir> -c gzip` (the CPU pass-through flow the reference runs, ref:snacc/cli.py:104-160):
    CSV bytes, phase banners and the fixed lines of the Markdown run log equal the reference CLI's."""
    from click.testing import CliRunner
    from snacc_amd.cli import cli
    g = golden["cli_gzip"]
    spec = golden["cli_lz4"]["sets"][g["set"]]
    d = materialise_cli_set(oracle_mod, spec, tmp_path / "fa")
    out = tmp_path / "gz.csv"
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SNACC_DEFLATE", "stdlib")        # the reference's own thread-pool flow, asked for explicitly
    res = CliRunner().invoke(cli, [str(d), "-o", str(out), "-c", "gzip", "-n", "2", "--no-show-progress"])
    assert res.exit_code == 0, res.output
    assert res.output == g["stdout"]
    assert out.read_text() == g["csv"].replace("{DIR}", str(d))
    log = (tmp_path / "gz.md").read_text().splitlines()
    fixed = [ln for ln in log if not ln.startswith(("* Analysis", "* Output", "* Python", "* snacc", "* py-lz4framed"))]
    assert fixed == [ln.replace("{DIR}", str(d)) for ln in g["log_fixed_lines"]]
    assert any(ln.startswith("* py-lz4framed: n/a (snacc_amd") for ln in log)


def test_cli_deprecated_flags_warn_and_work(tmp_path, monkeypatch):
    from click.testing import CliRunner
    from snacc_amd.cli import cli
    d = tmp_path / "in"
    d.mkdir()
    (d / "a.fa").write_text(">a\nACGTACGTAC\n")
    (d / "b.fa").write_text(">b\nACGTACGTTT\n")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SNACC_DEFLATE", "stdlib")
    res = CliRunner().invoke(cli, ["-d", str(d), "-f", str(d / "a.fa"), "-o", "o.csv", "-c", "zlib", "--no-show-progress", "--no-log"])
    assert res.exit_code == 0, res.output
    assert "deprecated" in res.output
    assert len((tmp_path / "o.csv").read_text().splitlines()) == 3


def test_cli_gzip_defaults_to_the_hip_backend_and_fails_loudly_without_it(tmp_path, monkeypatch):
    """`-c gzip` / `-c zlib` run on the HIP backend; without a device they fail, they do
    not quietly fall back to the CPU codec."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    from click.testing import CliRunner
    from snacc_amd.cli import cli
    d = tmp_path / "in"
    d.mkdir()
    (d / "a.fa").write_text(">a\nACGTACGTAC\n")
    (d / "b.fa").write_text(">b\nACGTACGTTT\n")
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("SNACC_DEFLATE", raising=False)
    res = CliRunner().invoke(cli, [str(d), "-o", "o.csv", "-c", "gzip", "--no-show-progress", "--no-log"])
    assert res.exit_code != 0
    assert not (tmp_path / "o.csv").exists()


def test_single_item_gzip_needs_the_hip_backend_by_default(sample_fa, monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    monkeypatch.delenv("SNACC_DEFLATE", raising=False)
    from snacc_amd.hip_backend import HipBackendError
    with pytest.raises(HipBackendError):
        compressed_size(sample_fa, "gzip")


def test_bench_reads_the_committed_profile_summaries():
    """bench.py quotes HBM traffic and the wave cycle account from the round's committed rocprofv3 / stats summaries
    (profiles/rNN_*.json, newest round first): the files parse, name their source and the commit they were taken at, and match
    the launch shape the bench line is quoted on."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for codec in ("lz4", "gzip", "zlib"):
        traffic, src = bench.pmc_traffic(84, 1024, 1000000, codec)
        assert traffic and traffic > 0 and src.startswith("profiles/r0") and "pmc_traffic" in src, (codec, traffic, src)
    assert bench.pmc_traffic(84, 1024, 1000000, "lz4")[1].startswith("profiles/r04_pmc_traffic.json @ ")
    assert "?" not in bench.pmc_traffic(84, 1024, 1000000, "lz4")[1]                  # names its commit
    assert bench.pmc_traffic(84, 1000, 1000000)[0] is None           # another launch shape: no figure
    # (round 4) the secondary data sets have counters of their own; a set measured on another launch shape has no figure
    tm, srcm = bench.pmc_traffic(84, 1024, 1000000, "lz4", "markov")
    assert tm and tm > 0 and srcm.startswith("profiles/r04_pmc_traffic_markov.json @ ")
    assert bench.pmc_traffic(84, 1024, 1000000, "lz4", "softmask5")[0] is None     # (taken at 83 rows: sets with other-case letters hold 83 chains)
    assert bench.pmc_traffic(83, 1024, 1000000, "lz4", "softmask5")[0] > 0
    assert bench.pmc_traffic(84, 1024, 1000000, "lz4", "no-such-set")[0] is None
    im = bench.issue_model()
    # (the two-lane loop of round 3: ~134 issue slots per chain-trip, 1.66 probes per chain-trip)
    assert 60 < im["issue_slots_per_trip"] < 160 and im["measured_cycles_per_trip"] > im["issue_slots_per_trip"] * 4
    assert 1.0 <= im["probes_per_chain_trip"] < 2.0
    acc = bench.cycle_account()
    assert 300 < acc["cycles_per_trip_in_loop"] < 800 and 0 < acc["share_outside_loop"] < 0.5 and "profiles/" in acc["source"]
