"""Multi-rank path on the gloo backend with CPU tensors, world sizes 2 and 3.

What runs here is the PRODUCT's code -- snacc_amd/distributed.py (work-balanced row blocks, padded
per-tile all-gathers, allgather_check as bench.py uses it) and snacc_amd/cli.gpu_matrix (every rank
uploads, rank 0 alone reports and returns the matrix, the group it started is destroyed) -- with a
checker-provided size provider (the oracle) standing in for the HIP context.  RCCL itself cannot be
exercised without a multi-GPU node."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _seqs(n, length):
    import oracle
    return [oracle.lcg_genome(100 + i, length + 997 * (i % 4) * (i + 1)) for i in range(n)]


def _init(rank, world, port):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"], os.environ["RANK"], os.environ["LOCAL_RANK"] = str(world), str(rank), "0"


def _worker_sharded(rank, world, port, n, length, outdir, weighted, tile_rows):
    _init(rank, world, port)
    import torch
    import torch.distributed as dist
    from oracle.loader import pairs_mt
    from snacc_amd.distributed import (all_pairs_sharded, allgather_check, gather_tile, init_process_group,
                                       lz4_row_weights)
    init_process_group("gloo")
    seqs = _seqs(n, length)
    calls = []

    def rows_fn(r0, r1):
        calls.append((r0, r1))
        return pairs_mt(seqs, r0, r1, 2)

    asked = []

    def singles_fn(r0, r1):                    # phase A of the rank's OWN rows only; the sizes ride on the tile gathers
        import oracle
        asked.append((r0, r1))
        return np.array([oracle.lz4f_size(s) for s in seqs[r0:r1]], dtype=np.uint32)

    weights = lz4_row_weights([len(s) for s in seqs]) if weighted else None
    full = all_pairs_sharded(n, rows_fn, weights=weights, tile_rows=tile_rows)
    full2, singles = all_pairs_sharded(n, rows_fn, weights=weights, tile_rows=tile_rows, singles_fn=singles_fn)
    assert np.array_equal(full, full2)
    calls[:] = calls[: len(calls) // 2]
    np.save(os.path.join(outdir, f"singles_{rank}.npy"), singles)
    np.save(os.path.join(outdir, f"asked_{rank}.npy"), np.array(asked, dtype=np.int64).reshape(-1, 2))
    np.save(os.path.join(outdir, f"full_{rank}.npy"), full)
    np.save(os.path.join(outdir, f"rows_{rank}.npy"), np.array(calls, dtype=np.int64).reshape(-1, 2))
    # bench.py's step: one tile per rank, gathered, checked on every rank
    tile = torch.full((3, n), rank + 1, dtype=torch.int32)
    gathered, _ = gather_tile(tile, world)
    ok = allgather_check(gathered, tile, rank, world)
    bad = gathered.clone()
    if rank == world - 1:
        bad[rank * 3, 0] += 1                                        # ONE rank sees a wrong slot: every rank must learn it
    bad_ok = allgather_check(bad, tile, rank, world)
    np.save(os.path.join(outdir, f"check_{rank}.npy"), np.array([ok, bad_ok, int(gathered[:, 0].sum())]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,weighted,tile_rows", [(2, 5, False, None), (2, 6, True, 2), (2, 1, False, None),
                                                        (3, 7, True, 1), (3, 8, False, 2), (3, 2, True, None)])
def test_allgather_assembly_matches_single_process(tmp_path, world, n, weighted, tile_rows):
    from oracle.loader import pairs_mt
    from snacc_amd.distributed import lz4_row_weights, shard_rows, shard_rows_weighted
    length = 3000
    port = 29500 + (os.getpid() * 7 + n * 13 + world) % 2000
    mp.spawn(_worker_sharded, args=(world, port, n, length, str(tmp_path), weighted, tile_rows), nprocs=world, join=True)
    seqs = _seqs(n, length)
    want = pairs_mt(seqs, 0, n, 2)
    blocks = (shard_rows_weighted(lz4_row_weights([len(s) for s in seqs]), world) if weighted
              else [shard_rows(n, world, r)[:2] for r in range(world)])
    covered = []
    import oracle
    want_singles = np.array([oracle.lz4f_size(s) for s in seqs], dtype=np.uint32)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"full_{r}.npy"), want)
        # every rank holds all N single sizes although it computed only those of its own block
        assert np.array_equal(np.load(tmp_path / f"singles_{r}.npy"), want_singles)
        asked = np.load(tmp_path / f"asked_{r}.npy").tolist()
        assert asked == ([list(blocks[r])] if blocks[r][1] > blocks[r][0] else [])
        rows = np.load(tmp_path / f"rows_{r}.npy")
        for a, b in rows:
            assert blocks[r][0] <= a < b <= blocks[r][1]
            covered += list(range(a, b))
        ok, bad_ok, colsum = np.load(tmp_path / f"check_{r}.npy")
        assert ok == 1 and bad_ok == 0 and colsum == 3 * sum(range(1, world + 1))
    assert sorted(covered) == list(range(n))


def test_shard_rows_cover_everything():
    from snacc_amd.distributed import shard_rows
    for n in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                r0, r1, per = shard_rows(n, world, r)
                assert r1 - r0 <= per
                got += list(range(r0, r1))
            assert got == list(range(n))


def test_weighted_shards_balance_ragged_lengths():
    """rows are cut by work, not by count.  On ragged genome lengths every rank's share of the
    lz4 work stays within one row of the ideal; equal counts do not."""
    from snacc_amd.distributed import lz4_row_weights, shard_rows, shard_rows_weighted
    rng = np.random.default_rng(3)
    for world in (2, 3, 8):
        lengths = np.sort(rng.integers(500_000, 1_500_000, 1024))          # sorted by size: file order often is
        w = lz4_row_weights(lengths)
        blocks = shard_rows_weighted(w, world)
        assert blocks[0][0] == 0 and blocks[-1][1] == 1024 and all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        share = np.array([w[a:b].sum() for a, b in blocks])
        assert share.max() <= w.sum() / world + w.max()
        # a codec without prefix reuse costs len_i * sum_j len_j per row: there equal counts are far off
        w2 = lengths.astype(np.float64) * lengths.sum()
        blocks2 = shard_rows_weighted(w2, world)
        share2 = np.array([w2[a:b].sum() for a, b in blocks2])
        by_count = np.array([w2[slice(*shard_rows(1024, world, r)[:2])].sum() for r in range(world)])
        assert share2.max() <= w2.sum() / world + w2.max() < by_count.max()
    assert shard_rows_weighted([], 3) == [(0, 0)] * 3
    assert shard_rows_weighted([0, 0, 0, 0], 2) == [(0, 2), (2, 4)]
    assert shard_rows_weighted([5.0], 3)[-1][1] == 1


def _worker_ecoli(rank, world, port, outdir):
    _init(rank, world, port)
    import ecoli_like as ec
    import oracle
    import torch.distributed as dist
    from oracle.loader import pairs_mt
    from snacc_amd.distributed import all_pairs_sharded, init_process_group, lz4_row_weights
    init_process_group("gloo")
    seqs = [ec.expected_sequence(ec.records(i, ec.make_genome(oracle, i, scale=1000)), True) for i in range(ec.N_GENOMES)]
    calls = []

    def rows_fn(r0, r1):
        calls.append((r0, r1))
        return pairs_mt(seqs, r0, r1, 2)

    # the row weights are those of the FULL-SIZE set (what a rank of the real run computes from ctx.lengths())
    full = all_pairs_sharded(ec.N_GENOMES, rows_fn, weights=lz4_row_weights(ec.lengths()), tile_rows=16)
    np.save(os.path.join(outdir, f"full_{rank}.npy"), full)
    np.save(os.path.join(outdir, f"rows_{rank}.npy"), np.array(calls, dtype=np.int64).reshape(-1, 2))
    dist.destroy_process_group()


def test_config5_set_through_the_sharded_split_world3(tmp_path):
    """BASELINE.json configs[4] (92 genomes, row-sharded): the 92 lengths of the stand-in set (tests/ecoli_like.py)
    cut into work-balanced row blocks for 3 ranks, every rank's tiles all-gathered (gloo), the assembled matrix
    equal to the single-process one on every rank.  Sizes come from the set at 1/1000 scale (the oracle at full
    size is the GPU test's job); the weights are those of the full-size lengths."""
    import ecoli_like as ec
    import oracle
    from oracle.loader import pairs_mt
    from snacc_amd.distributed import lz4_row_weights, shard_rows_weighted
    world, n = 3, ec.N_GENOMES
    port = 33500 + (os.getpid() * 5) % 2000
    mp.spawn(_worker_ecoli, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    w = lz4_row_weights(ec.lengths())
    blocks = shard_rows_weighted(w, world)
    assert blocks[0][0] == 0 and blocks[-1][1] == n and all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
    share = np.array([w[a:b].sum() for a, b in blocks])
    assert share.max() <= w.sum() / world + w.max()                      # no rank more than one row above the ideal share
    assert share.max() / share.min() < 1.10
    seqs = [ec.expected_sequence(ec.records(i, ec.make_genome(oracle, i, scale=1000)), True) for i in range(n)]
    want = pairs_mt(seqs, 0, n, 2)
    covered = []
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"full_{r}.npy"), want)
        for a, b in np.load(tmp_path / f"rows_{r}.npy"):
            assert blocks[r][0] <= a < b <= blocks[r][1] and b - a <= 16
            covered += list(range(a, b))
    assert sorted(covered) == list(range(n))


def test_config5_set_files_read_natively_equal_the_python_statement(tmp_path):
    """The stand-in set's FASTA files (multi-record, N runs, IUPAC codes, a lower-case stretch) through the native
    reader with per-record reverse complement = the Python statement of ref:snacc/pairwise_ncd.py:29-36 (1/100 scale)."""
    import ecoli_like as ec
    import oracle
    from snacc_amd import hip_backend
    for i in (0, 7, 33, 35, 70, 91):
        recs = ec.records(i, ec.make_genome(oracle, i, scale=100))
        p = tmp_path / f"g{i:02d}.fna"
        ec.write_fasta_fast(p, recs)
        for rc in (False, True):
            assert hip_backend.fasta_extract(p, rc) == bytes(ec.expected_sequence(recs, rc)), (i, rc)


class _CheckerContext:
    """Stands in for HipContext in cli.gpu_matrix: sizes from the oracle, written where the caller says."""
    def __init__(self, device):
        import torch
        self.device = device
        self.torch_device = torch.device("cpu")
        self.seqs = []
        self.closed = False
        self.options = {}
        self.singles_asked = []

    def set_option(self, key, value):
        self.options[key] = value

    def singles_rows(self, r0, r1):
        import oracle
        self.singles_asked.append((r0, r1))
        return np.array([oracle.lz4f_size(s) for s in self.seqs[r0:r1]], dtype=np.uint32)

    arena_limit = None           # residues one upload may hold (tests of the blocked path set it)

    def upload_fasta(self, paths, reverse_complement=False):
        from snacc_amd import fasta
        from snacc_amd.hip_backend import ArenaTooBig
        seqs = [np.frombuffer(fasta.read_sequence(p, reverse_complement).encode(), dtype=np.uint8) for p in paths]
        if self.arena_limit is not None and sum(len(s) + 128 for s in seqs) >= self.arena_limit:
            raise ArenaTooBig("ASCII arena exceeds the offset range of one upload")
        self.seqs = seqs
        self.n = len(self.seqs)
        self.uploads = getattr(self, "uploads", 0) + 1

    def pairs_list(self, ij):
        from oracle.loader import pairs_list_mt
        return pairs_list_mt(self.seqs, np.array(ij, dtype=np.int32), 2)

    def lengths(self):
        return np.array([len(s) for s in self.seqs], dtype=np.uint64)

    def singles(self):
        import oracle
        return np.array([oracle.lz4f_size(s) for s in self.seqs], dtype=np.uint32)

    def pairs_device(self, r0, r1, ptr, stream):
        from oracle.loader import pairs_mt
        a = np.ascontiguousarray(pairs_mt(self.seqs, r0, r1, 2))
        ctypes.memmove(ptr, a.ctypes.data, a.nbytes)

    def sync(self, stream):
        pass

    def close(self):
        self.closed = True


def _worker_cli(rank, world, port, fadir, outdir, arena_limit=None):
    _init(rank, world, port)
    import torch.distributed as dist
    from click.testing import CliRunner
    from snacc_amd import cli as cli_mod
    made = []
    if arena_limit:
        os.environ["SNACC_ARENA_LIMIT"] = str(arena_limit)

    def factory(dev):
        made.append(_CheckerContext(dev))
        made[-1].arena_limit = arena_limit
        return made[-1]

    real = cli_mod.gpu_matrix
    cli_mod.gpu_matrix = lambda *a, **k: real(*a, ctx_factory=factory, backend="gloo", **k)
    out = Path(outdir) / f"out_{rank}.csv"
    os.chdir(outdir)
    res = CliRunner().invoke(cli_mod.cli, [fadir, "-o", str(out), "-c", "lz4", "--no-show-progress", "--no-log"])
    Path(outdir, f"cli_{rank}.txt").write_text(f"{res.exit_code}\n{int(out.exists())}\n{int(dist.is_initialized())}\n"
                                               f"{int(made[0].closed)}\n{res.output}\n{res.exception!r}")
    Path(outdir, f"uploads_{rank}.txt").write_text(str(getattr(made[0], "uploads", 0)))
    Path(outdir, f"asked_{rank}.txt").write_text(repr((made[0].options, made[0].singles_asked)))


@pytest.mark.parametrize("world", [2, 3])
def test_cli_multi_rank_only_rank0_reports_and_writes(tmp_path, world):
    import oracle
    from conftest import write_fasta
    from snacc_amd.matrix import ncd_matrix
    fadir = tmp_path / "fa"
    fadir.mkdir()
    seqs = _seqs(5, 20000)
    for k, sq in enumerate(seqs):
        write_fasta(fadir / f"g{k}.fasta", [("r", bytes(sq).decode())])
    port = 31500 + (os.getpid() * 3 + world) % 2000
    mp.spawn(_worker_cli, args=(world, port, str(fadir), str(tmp_path)), nprocs=world, join=True)
    singles = np.array([oracle.lz4f_size(s) for s in seqs], dtype=np.int64) + 33
    pairs = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.int64) + 33
    want = ncd_matrix(singles, pairs)
    for r in range(world):
        code, wrote, still_init, closed, *output = (tmp_path / f"cli_{r}.txt").read_text().split("\n")
        assert code == "0" and still_init == "0" and closed == "1", (r, output)
        assert wrote == ("1" if r == 0 else "0")
        assert ("Compressing pairs..." in "\n".join(output)) == (r == 0)
    got = np.loadtxt(tmp_path / "out_0.csv", delimiter=",", skiprows=1, usecols=range(1, 6))
    assert np.array_equal(got, want)
    # phase A per owner: every rank deferred it at upload and asked for the singles of its own block of rows, once
    from snacc_amd.distributed import lz4_row_weights, shard_rows_weighted
    blocks = shard_rows_weighted(lz4_row_weights([len(s) for s in seqs]), world)
    for r in range(world):
        options, asked = eval((tmp_path / f"asked_{r}.txt").read_text())
        assert options.get("defer_singles") == 1
        assert asked == ([tuple(blocks[r])] if blocks[r][1] > blocks[r][0] else [])


@pytest.mark.parametrize("world", [2, 3])
def test_cli_multi_rank_set_beyond_one_upload_is_sharded_by_group_pairs(tmp_path, world):
    """A set that does not fit one upload (ArenaTooBig) under torchrun: the group pairs of
    snacc_amd.cli.blocked_sizes are dealt over the ranks (no rank raises, no rank does all the uploads), the two
    result arrays are summed over the ranks and rank 0 writes the same CSV as a single upload gives."""
    import oracle
    from conftest import write_fasta
    from snacc_amd.matrix import ncd_matrix
    fadir = tmp_path / "fa"
    fadir.mkdir()
    seqs = _seqs(7, 20000)
    for k, sq in enumerate(seqs):
        write_fasta(fadir / f"g{k}.fasta", [("r", bytes(sq).decode())])
    port = 35500 + (os.getpid() * 3 + world) % 2000
    # 7 files of ~21-27 kB against a limit of 110 kB: groups of two files (45 % of the limit), 10 group pairs
    mp.spawn(_worker_cli, args=(world, port, str(fadir), str(tmp_path), 110_000), nprocs=world, join=True)
    singles = np.array([oracle.lz4f_size(s) for s in seqs], dtype=np.int64) + 33
    pairs = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.int64) + 33
    want = ncd_matrix(singles, pairs)
    uploads = []
    for r in range(world):
        code, wrote, still_init, closed, *output = (tmp_path / f"cli_{r}.txt").read_text().split("\n")
        assert code == "0" and still_init == "0" and closed == "1", (r, output)
        assert wrote == ("1" if r == 0 else "0")
        uploads.append(int((tmp_path / f"uploads_{r}.txt").read_text()))
    assert sum(uploads) >= 10 and max(uploads) < sum(uploads)           # every rank took a share of the group pairs
    got = np.loadtxt(tmp_path / "out_0.csv", delimiter=",", skiprows=1, usecols=range(1, 8))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("world", [2])
def test_cli_multi_rank_pair_that_fits_no_upload_ends_every_rank(tmp_path, world):
    """A single pair that fits no upload is met by ONE rank of the blocked path (the one dealt that group pair).  The ranks
    agree on the failure before the data reduce: every rank ends with a non-zero status and a message, none blocks in the
    collective, nothing is written."""
    from conftest import write_fasta
    fadir = tmp_path / "fa"
    fadir.mkdir()
    import oracle
    # two files of 9 kB whose pair exceeds the 15 kB one upload holds, two small ones: groups {0}, {1}, {2, 3}; of the six
    # group pairs only (0, 1) fails, and the round-robin deal gives it to rank 1 alone
    for k, length in enumerate((9000, 9000, 1500, 1500)):
        write_fasta(fadir / f"g{k}.fasta", [("r", bytes(oracle.lcg_genome(300 + k, length)).decode())])
    port = 37500 + (os.getpid() * 3 + world) % 2000
    mp.spawn(_worker_cli, args=(world, port, str(fadir), str(tmp_path), 15_000), nprocs=world, join=True)
    texts = []
    for r in range(world):
        code, wrote, *output = (tmp_path / f"cli_{r}.txt").read_text().split("\n")
        assert code != "0" and wrote == "0", (r, output)
        texts.append("\n".join(output))
    assert "do not fit one upload" in texts[1] and "do not fit one upload" not in texts[0]
    assert "another rank could not compute" in texts[0]


def test_blocked_sizes_splits_an_upload_that_still_does_not_fit(tmp_path):
    """Groups are sized from file sizes; when an upload of two groups still exceeds the limit (arena padding under a
    small limit) it is split in halves, and a single pair that cannot fit ends with a message, not a traceback."""
    import click
    import oracle
    from conftest import write_fasta
    from snacc_amd.cli import blocked_sizes
    seqs = _seqs(6, 9000)
    files = []
    for k, sq in enumerate(seqs):
        files.append(tmp_path / f"g{k}.fasta")
        write_fasta(files[-1], [("r", bytes(sq).decode())])
    ctx = _CheckerContext(0)
    ctx.arena_limit = 37_000                                               # any two files fit, the four of two groups do not
    os.environ["SNACC_ARENA_LIMIT"] = "70000"                               # ... but the groups are cut for 31 kB of file each
    try:
        singles, pairs = blocked_sizes(ctx, files, "lz4", False)
    finally:
        del os.environ["SNACC_ARENA_LIMIT"]
    assert singles.tolist() == [oracle.lz4f_size(s) + 33 for s in seqs]
    assert pairs.tolist() == [[oracle.lz4f_size_pair(a, b) + 33 for b in seqs] for a in seqs]
    ctx.arena_limit = 15_000                                               # not even one pair fits
    with pytest.raises(click.ClickException, match="do not fit one upload"):
        blocked_sizes(ctx, files, "lz4", False)


def test_failed_rendezvous_exits_non_zero(tmp_path):
    """A rank that cannot join the group ends with status 3 and says why (no silent single-rank result)."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='1', WORLD_SIZE='2', RANK='1')\n"
            "from snacc_amd.distributed import init_process_group\n"
            "import datetime, torch.distributed as dist\n"
            "real = dist.init_process_group\n"
            "dist.init_process_group = lambda *a, **k: real(*a, timeout=datetime.timedelta(seconds=2), **k)\n"
            "init_process_group('gloo')\n" % str(ROOT))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 3
    assert "init_process_group(gloo) failed" in res.stderr
