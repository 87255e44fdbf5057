"""GPU parity tests of the gzip / zlib path: the HIP backend, through the C-ABI,
against the codec the reference calls (the interpreter's gzip / zlib, ref:snacc/pairwise_ncd.py:73-78)
and against the deflate oracle.  Bit-exact sizes."""
import gzip
import zlib

import numpy as np
import pytest

from conftest import materialise_cli_set

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
CODEC = {"gzip": gzip.compress, "zlib": zlib.compress}


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from snacc_amd import hip_backend
    hip_backend.load()
    return hip_backend


def _b(x):
    return x if isinstance(x, (bytes, bytearray)) else bytes(np.ascontiguousarray(x, dtype=np.uint8))


def _check_all_vs_codec(hip, seqs):
    """singles and every ordered pair, both algorithms, against the real codec"""
    raw = [_b(s) for s in seqs]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        for alg, fn in CODEC.items():
            s, p = ctx.deflate_singles(alg), ctx.deflate_pairs(alg)
            es = np.array([len(fn(a)) for a in raw], dtype=np.uint32)
            ep = np.array([[len(fn(a + b)) for b in raw] for a in raw], dtype=np.uint32)
            assert np.array_equal(s, es), (alg, np.flatnonzero(s != es))
            assert np.array_equal(p, ep), (alg, np.argwhere(p != ep)[:5])


def test_tiny_and_empty_inputs(hip):
    _check_all_vs_codec(hip, [b"ACGT" * 10, b"ACGTTGCA" * 3, b"A", b"", b"ACGTN" * 5, b"GATTACA" * 1000, b"AC", b"ACG",
                              b"ACGTACGTACGTA", bytes(range(256))])


def test_lengths_around_window_slides_and_block_cuts(hip, oracle_mod):
    lens = [32767, 32768, 32769, 65274, 65275, 65536, 65537, 65536 + 262, 98304, 131073, 600, 601, 33000]
    _check_all_vs_codec(hip, [oracle_mod.lcg_genome(100 + i, n) for i, n in enumerate(lens)])


def test_mixed_alphabets_stored_blocks_and_long_runs(hip, oracle_mod):
    rng = np.random.default_rng(11)
    seqs = [rng.integers(0, 256, 90000, dtype=np.uint8),                       # incompressible: stored blocks
            rng.choice(ACGT, 120000),
            np.tile(rng.choice(ACGT, 700), 200),                                 # period 700: matches of 258
            np.full(100000, ord("A"), dtype=np.uint8),                           # one long run
            rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), 80000),
            np.repeat(rng.choice(ACGT, 3000), 40)[:100000].copy(),               # homopolymer runs
            np.concatenate([rng.integers(0, 256, 40000, dtype=np.uint8), rng.choice(ACGT, 70000)]),
            rng.choice(np.frombuffer(b"AC", dtype=np.uint8), 70000)]
    _check_all_vs_codec(hip, seqs)


def test_related_genomes_and_tandem_repeats(hip, oracle_mod):
    rng = np.random.default_rng(12)
    x = oracle_mod.lcg_genome(7, 150000)
    y = x.copy()
    hit = rng.random(len(y)) < 0.02
    y[hit] = rng.choice(ACGT, int(hit.sum()))
    unit = rng.choice(ACGT, 1900)
    t = np.tile(unit, 80)[:140000].copy()
    hit = rng.random(len(t)) < 0.01
    t[hit] = rng.choice(ACGT, int(hit.sum()))
    _check_all_vs_codec(hip, [x, y, x[5000:], t, t[::-1].copy(), y[:70000]])


def test_segmented_and_serial_sequence_passes_agree(hip, oracle_mod):
    """Every single sequence is parsed in parallel 32 KiB segments that are stitched where two parsers
    provably agree; `deflate_serial=1` parses it with one wavefront from start to end.  Same sizes, for
    random genomes (many segments), periodic data (parsers that may never meet: serial fallback) and runs."""
    rng = np.random.default_rng(21)
    t = np.tile(rng.choice(ACGT, 1900), 140)[:250000].copy()
    hit = rng.random(len(t)) < 0.01
    t[hit] = rng.choice(ACGT, int(hit.sum()))
    seqs = [oracle_mod.lcg_genome(70, 300001), oracle_mod.lcg_genome(71, 2500000), np.tile(oracle_mod.lcg_genome(72, 1000), 400),
            np.full(200000, ord("A"), dtype=np.uint8), t, rng.integers(0, 256, 150000, dtype=np.uint8),
            np.tile(np.frombuffer(b"ACGTTGCA", dtype=np.uint8), 30000)]
    raw = [_b(x) for x in seqs]
    with hip.HipContext(0) as a, hip.HipContext(0, deflate_serial=1) as b:
        a.upload(seqs)
        b.upload(seqs)
        for alg, fn in CODEC.items():
            sa, sb = a.deflate_singles(alg), b.deflate_singles(alg)
            assert np.array_equal(sa, sb), alg
            assert [int(v) for v in sa] == [len(fn(x)) for x in raw], alg
            pa, pb = a.deflate_pairs(alg), b.deflate_pairs(alg)
            assert np.array_equal(pa, pb), alg
            for i, j in ((0, 1), (1, 0), (2, 3), (3, 2), (4, 4), (5, 6), (6, 0)):
                assert int(pa[i, j]) == len(fn(raw[i] + raw[j])), (alg, i, j)


def test_segments_on_data_with_rare_matches(hip, oracle_mod):
    """Regression (found by tools/gpu_deflate_fuzz.py, seeds 20016 and 20508): a segment job stops behind a match; on
    data with hardly any match (stretches of random bytes) it ran on and overflowed its scratch stream into the
    next segment's, and the block that is open at the restart point got wrong symbol counts.  The two sequence
    sets of those seeds, singles and all pairs, against the codec."""
    from fuzzgen import make_set
    for seed, repeats in ((20016, 5), (20508, 2)):       # the overflow raced with the neighbour's own writes: repeat
        seqs = make_set(seed)
        for _ in range(repeats):
            _check_all_vs_codec(hip, seqs)


def test_six_byte_index_on_and_off_agree(hip, oracle_mod):
    """gzip's match search first tries the chain members that share six bytes with the probe (second index);
    `deflate_kmer=0` walks every chain in full.  Same sizes -- on genomes, relatives, low-complexity and
    protein-like data, and across the seam (pairs)."""
    rng = np.random.default_rng(31)
    x = oracle_mod.lcg_genome(80, 180000)
    y = x.copy()
    hit = rng.random(len(y)) < 0.01
    y[hit] = rng.choice(ACGT, int(hit.sum()))
    seqs = [x, y[3000:], oracle_mod.lcg_genome(81, 90000), rng.choice(np.frombuffer(b"AC", dtype=np.uint8), 60000),
            rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), 70000),
            np.repeat(rng.choice(ACGT, 4000), 30)[:100000].copy(), x[:70], x[:6], x[:5]]
    with hip.HipContext(0) as a, hip.HipContext(0, deflate_kmer=0) as b:
        a.upload(seqs)
        b.upload(seqs)
        assert np.array_equal(a.deflate_singles("gzip"), b.deflate_singles("gzip"))
        pa, pb = a.deflate_pairs("gzip"), b.deflate_pairs("gzip")
        assert np.array_equal(pa, pb)
    raw = [_b(s) for s in seqs]
    for i, j in ((0, 1), (1, 0), (2, 3), (5, 0), (6, 7), (8, 6), (0, 8)):
        assert int(pa[i, j]) == len(gzip.compress(raw[i] + raw[j])), (i, j)


def test_pairs_with_and_without_the_x_prefix_restart_agree(hip, oracle_mod):
    """A pair job normally restarts from x's stored stream ~600 bytes before the seam (and falls back to parsing x
    from its first byte when a stored-block decision of x could depend on the stream length).  Both ways, same sizes."""
    rng = np.random.default_rng(41)
    seqs = [oracle_mod.lcg_genome(85, 150000), oracle_mod.lcg_genome(86, 40000), rng.integers(0, 256, 70000, dtype=np.uint8),
            np.concatenate([rng.integers(0, 256, 30000, dtype=np.uint8), oracle_mod.lcg_genome(87, 60000)]),
            np.tile(oracle_mod.lcg_genome(88, 3000), 30)]
    with hip.HipContext(0) as a, hip.HipContext(0, deflate_norestart=1) as b:
        a.upload(seqs)
        b.upload(seqs)
        for alg in ("gzip", "zlib"):
            pa, pb = a.deflate_pairs(alg), b.deflate_pairs(alg)
            assert np.array_equal(pa, pb), (alg, np.argwhere(pa != pb)[:4])
    raw = [_b(s) for s in seqs]
    assert int(pb[0, 2]) == len(zlib.compress(raw[0] + raw[2])) and int(pb[3, 1]) == len(zlib.compress(raw[3] + raw[1]))


def test_device_output_rows_equal_host_rows(hip, oracle_mod):
    import torch
    seqs = [oracle_mod.lcg_genome(120 + i, 60000 + 9000 * i) for i in range(6)]
    dev = torch.device("cuda", 0)
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        for alg, wrapper in (("gzip", 18), ("zlib", 6)):
            host = ctx.deflate_pairs(alg, 1, 5)
            out = torch.zeros((4, 6), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.deflate_pairs_device(alg, 1, 5, out.data_ptr())
            ctx.sync()
            assert ctx.deflate_last_ms() > 0
            assert np.array_equal(out.cpu().numpy().view(np.uint32) + np.uint32(wrapper), host)
            stream = torch.cuda.Stream(dev)
            out2 = torch.zeros((6, 6), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.deflate_pairs_device(alg, 0, 6, out2.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            assert np.array_equal(out2.cpu().numpy().view(np.uint32)[1:5] + np.uint32(wrapper), host)
    from snacc_amd.distributed import all_pairs_deflate_hip
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        assert np.array_equal(all_pairs_deflate_hip(ctx, 6, "zlib")[1:5], host)        # world size 1: no collective


def test_argument_and_state_errors(hip):
    with hip.HipContext(0) as ctx:
        with pytest.raises(hip.HipBackendError):
            ctx.deflate_singles("gzip")                     # nothing uploaded
        ctx.upload([b"ACGT" * 100, b"GATTACA" * 50])
        out = np.zeros(2, dtype=np.uint32)
        assert ctx._L.snk_deflate_singles(ctx._h, 5, out.ctypes.data) == -1       # SNK_E_ARG: only levels 9 and 6
        assert ctx._L.snk_deflate_pairs(ctx._h, 9, 0, 3, out.ctypes.data) == -1   # row range out of bounds
        bad = np.array([[0, 2]], dtype=np.int32)
        assert ctx._L.snk_deflate_pairs_list(ctx._h, 9, 1, bad.ctypes.data, out.ctypes.data) == -1
        assert [int(v) for v in ctx.deflate_singles("gzip")] == [len(gzip.compress(b"ACGT" * 100)), len(gzip.compress(b"GATTACA" * 50))]


def test_matches_running_into_the_end_of_a_64k_stream(hip, oracle_mod):
    """Streams of 65 275 .. 65 536 bytes end with zlib's window slid once, so a match compare that runs past the
    last byte sees the data 32 KiB earlier, not zeros (CPU: test_bytes_behind_the_end_of_the_input).  Inputs whose
    tail occurs twice before -- the older copy continues like those remnant bytes, the newer one does not --
    as single sequences and as pairs whose total length falls into that range.  (zlib clamps nice_match to the
    bytes left, so the first chain member that reaches the end wins either way: this guards the lengths and the
    end-of-stream handling around that range rather than the remnant rule itself.)"""
    rng = np.random.default_rng(65275)
    seqs = []
    for n in (65275, 65300, 65400, 65500, 65536, 65274, 65537):
        a = rng.choice(ACGT, n)
        for tl in (12, 40):
            tail = a[n - tl:].copy()
            rem = a[n - 32768:n - 32768 + 200].copy()            # what the window holds behind the end after the slide
            q2, q1 = n - 20000 - tl, n - 5000 - tl
            b = a.copy()
            b[q2:q2 + tl] = tail
            b[q2 + tl:q2 + tl + 200] = rem                        # the older copy goes on like the remnant
            b[q1:q1 + tl] = tail
            b[q1 + tl] = ACGT[(int(np.flatnonzero(ACGT == rem[0])[0]) + 1) % 4]   # the newer one does not
            seqs.append(b)
    seqs += [seqs[0][:30000], seqs[3][30000:]]                    # 30 000 + 35 500 = 65 500 as a pair
    _check_all_vs_codec(hip, seqs[:6] + seqs[-2:])
    raw = [_b(s) for s in seqs]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        for alg, fn in CODEC.items():
            assert [int(v) for v in ctx.deflate_singles(alg)] == [len(fn(r)) for r in raw], alg


def test_pair_lists_tiles_and_single_items(hip, oracle_mod):
    seqs = [oracle_mod.lcg_genome(60 + i, 66000 + 7777 * i) for i in range(7)]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        for alg in ("gzip", "zlib"):
            full = ctx.deflate_pairs(alg)
            tiles = np.concatenate([ctx.deflate_pairs(alg, 0, 2), ctx.deflate_pairs(alg, 2, 7)])
            assert np.array_equal(full, tiles)
            ij = [(i, j) for i in range(7) for j in range(7)][::-1]
            assert np.array_equal(ctx.deflate_pairs_list(alg, ij), full[::-1, ::-1].reshape(-1))
            assert np.array_equal(ctx.deflate_pairs_list(alg, [(i, -1) for i in range(7)]), ctx.deflate_singles(alg))


def test_second_upload_replaces_indexes_and_streams(hip, oracle_mod):
    a = [oracle_mod.lcg_genome(90 + i, 70000 + 900 * i) for i in range(4)]
    b = [oracle_mod.lcg_genome(95 + i, 40000 + 30000 * i) for i in range(3)]
    with hip.HipContext(0) as ctx:
        for seqs in (a, b, a[:2]):
            ctx.upload(seqs)
            raw = [_b(s) for s in seqs]
            for alg, fn in CODEC.items():
                assert [int(v) for v in ctx.deflate_singles(alg)] == [len(fn(r)) for r in raw]
                p = ctx.deflate_pairs(alg)
                assert int(p[1, 0]) == len(fn(raw[1] + raw[0])) and int(p[0, len(seqs) - 1]) == len(fn(raw[0] + raw[-1]))
            assert ctx.singles().shape == (len(seqs),)        # the lz4 side of the same context still works


def test_1mbp_pairs_sample_against_the_oracle(hip, oracle_mod):
    """BASELINE's 1 Mbp size: the full 12 x 12 matrices on the GPU, a sample of pairs against the CPU
    oracle (about 2 s per gzip pair there), plus properties that need no oracle."""
    from oracle import deflate as D
    seqs = [oracle_mod.lcg_genome(1 + i, 1000000) for i in range(12)]
    empty = np.zeros(0, dtype=np.uint8)
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs + [empty])
        for alg, fn in (("gzip", D.gzip_size), ("zlib", D.zlib_size)):
            s, p = ctx.deflate_singles(alg), ctx.deflate_pairs(alg)
            assert np.array_equal(p[:12, 12], s[:12]) and np.array_equal(p[12, :12], s[:12])   # x + "" == x == "" + x
            for i, j in ((0, 1), (1, 0), (3, 3), (11, 5)):
                assert int(p[i, j]) == fn(seqs[i], seqs[j]), (alg, i, j)
            assert int(s[4]) == fn(seqs[4])


def test_golden_deflate_sizes_through_cabi(hip, golden, oracle_mod):
    """The committed zlib-1.2.11 sizes (tests/golden/golden.json), singles and pairs up to 1 Mbp, through the C-ABI."""
    from conftest import lcg_bytes
    g = golden["deflate_sizes"]
    for row in g["cases"]:
        n = row["n"]
        x, y = oracle_mod.lcg_genome(1, n), oracle_mod.lcg_genome(2, n)
        z = oracle_mod.lcg_mutant(x, 3)
        with hip.HipContext(0) as ctx:
            ctx.upload([x, y, z])
            for alg in ("gzip", "zlib"):
                s = ctx.deflate_singles(alg)
                assert [int(v) for v in s] == [row[k][alg] for k in ("x", "y", "z")], (n, alg)
                p = ctx.deflate_pairs_list(alg, [(0, 1), (1, 0), (0, 0), (0, 2)])
                assert [int(v) for v in p] == [row[k][alg] for k in ("xy", "yx", "xx", "xz")], (n, alg)
    seqs = [oracle_mod.lcg_genome(r["seed"], r["n"]) for r in g["ragged"]]
    seqs += [lcg_bytes(r["seed"], r["n"], bytes.fromhex(r["alphabet_hex"])) for r in g["other_alphabets"]]
    exp = g["ragged"] + g["other_alphabets"]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        for alg in ("gzip", "zlib"):
            assert [int(v) for v in ctx.deflate_singles(alg)] == [r[alg] for r in exp], alg


def test_python_batched_api_and_ncd(hip, oracle_mod):
    from snacc_amd.pairwise_ncd import all_pairs, ncd_matrix_gpu, compute_distance
    seqs = [bytes(oracle_mod.lcg_genome(30 + i, 40000 + 5000 * i)) for i in range(5)]
    for alg, fn in CODEC.items():
        singles, pairs = all_pairs(seqs, alg)
        assert [int(v) for v in singles] == [len(fn(a)) + 33 for a in seqs]
        m = ncd_matrix_gpu(seqs, alg)
        for i in range(5):
            for j in range(5):
                exp = compute_distance(len(fn(seqs[i])) + 33, len(fn(seqs[j])) + 33,
                                       len(fn(seqs[i] + seqs[j])) + 33, len(fn(seqs[j] + seqs[i])) + 33)
                assert abs(m[i, j] - exp) <= 1e-6          # north_star's tolerance; equal integers give equal floats
                assert m[i, j] == exp


def test_python_api_single_items(hip, golden, monkeypatch):
    """compressed_size(path | (path, path), "gzip" | "zlib") -- the reference's granularity -- on the GPU."""
    from pathlib import Path
    from snacc_amd import compressed_size
    monkeypatch.delenv("SNACC_DEFLATE", raising=False)
    p = Path(__file__).parent / "golden" / "sample_deflate.fa"
    p.write_text(">derice\nACTGACTAGCTAGCTAACTG\n>sanka\nGCATCGTAGCTAGCTACGAT\n>junior\nCATCGATCGTACGTACGTAG\n>yul\nATCGATCGATCGTACGATCG\n")
    try:
        g = golden["sample_fa"]
        for alg in ("gzip", "zlib"):
            assert compressed_size(p, alg) == (p, g["sizes_single"][alg])
            assert compressed_size((p, p), alg) == ((p, p), g["sizes_selfpair"][alg])
    finally:
        p.unlink()


def test_cli_gzip_csv_equals_reference_cli(hip, golden, oracle_mod, tmp_path, monkeypatch):
    """`snacc <dir> -c gzip` on the HIP backend writes the CSV the reference CLI wrote (golden fixture)."""
    from click.testing import CliRunner
    from snacc_amd.cli import cli
    g = golden["cli_gzip"]
    d = materialise_cli_set(oracle_mod, golden["cli_lz4"]["sets"][g["set"]], tmp_path / "fa")
    out = tmp_path / "gz.csv"
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("SNACC_DEFLATE", raising=False)
    res = CliRunner().invoke(cli, [str(d), "-o", str(out), "-c", "gzip", "--no-show-progress"])
    assert res.exit_code == 0, res.output
    assert res.output == g["stdout"]
    assert out.read_text() == g["csv"].replace("{DIR}", str(d))


def test_cli_zlib_equals_thread_pool_flow(hip, golden, oracle_mod, tmp_path, monkeypatch):
    from click.testing import CliRunner
    from snacc_amd.cli import cli
    d = materialise_cli_set(oracle_mod, golden["cli_lz4"]["sets"]["acgt_small"], tmp_path / "fa")
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("SNACC_DEFLATE", raising=False)
    a = CliRunner().invoke(cli, [str(d), "-o", "gpu.csv", "-c", "zlib", "-r", "--no-show-progress", "--no-log"])
    monkeypatch.setenv("SNACC_DEFLATE", "stdlib")
    b = CliRunner().invoke(cli, [str(d), "-o", "cpu.csv", "-c", "zlib", "-r", "-n", "4", "--no-show-progress", "--no-log"])
    assert a.exit_code == 0 and b.exit_code == 0, (a.output, b.output)
    assert (tmp_path / "gpu.csv").read_text() == (tmp_path / "cpu.csv").read_text()
