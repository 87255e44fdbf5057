"""Multi-rank assembly path on the gloo backend, world_size 2, CPU tensors.
The per-rank size provider is the oracle (the checker); the sharding, padding and all-gather
code is the product's (snacc_amd/distributed.py)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, n, length, outdir):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    import oracle
    from oracle.loader import pairs_mt
    from snacc_amd.distributed import all_pairs_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = [oracle.lcg_genome(100 + i, length + 37 * i) for i in range(n)]
    calls = []

    def rows_fn(r0, r1):
        calls.append((r0, r1))
        return pairs_mt(seqs, r0, r1, 2) if r1 > r0 else np.zeros((0, n), dtype=np.uint32)

    full = all_pairs_sharded(n, rows_fn)
    np.save(os.path.join(outdir, f"full_{rank}.npy"), full)
    np.save(os.path.join(outdir, f"rows_{rank}.npy"), np.array(calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 6, 1])
def test_two_rank_allgather_matches_single_process(tmp_path, n):
    import oracle
    from oracle.loader import pairs_mt
    length, world = 3000, 2
    port = 29500 + (os.getpid() + n) % 2000
    mp.spawn(_worker, args=(world, port, n, length, str(tmp_path)), nprocs=world, join=True)
    seqs = [oracle.lcg_genome(100 + i, length + 37 * i) for i in range(n)]
    want = pairs_mt(seqs, 0, n, 2)
    rows = []
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"full_{r}.npy"), want)
        rows.append(tuple(np.load(tmp_path / f"rows_{r}.npy")[0]))
    per = (n + world - 1) // world
    assert rows == [(0, min(per, n)), (min(per, n), n)]


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
