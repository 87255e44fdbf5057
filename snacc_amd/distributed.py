"""Row-sharded all-pairs over the GPUs of one node.

Every rank holds all sequences (1024 x 1 Mbp is 1 GB of ASCII, 256 MB packed -- nothing
against 288 GB of HBM) and computes the ordered pairs (i, j) of a contiguous block of prefix
rows i.  Blocks are cut so that every rank gets about the same WORK (:func:`shard_rows_weighted`;
for lz4 a pair (i, j) costs about ``tail(x_i) + len(y_j)`` because whole 64 KiB blocks of x come
from its prefix snapshot, for gzip / zlib about ``len(y_j)``).  There is no communication inside
the loop; the only exchange is the all-gather of the u32 size tiles (RCCL over xGMI on GPUs: 4 MB
in total at N = 1024, latency-bound), issued per row tile on a second stream so that the gather of
tile k runs under the kernels of tile k+1.  NCD floats are computed on the host afterwards.

The same code runs on the ``gloo`` backend with CPU tensors; the CPU test-suite drives it with a
checker-provided size provider at world sizes 2 and 3 (tests/test_distributed_gloo.py).  It has
NOT run on RCCL yet: no multi-GPU node has been available to the build.
"""
import sys

import numpy as np


def shard_rows(n, world, rank):
    """Contiguous row block of `rank` by COUNT: (r0, r1, rows_per_rank), rows_per_rank = ceil(n / world)."""
    per = (n + world - 1) // world
    r0 = min(rank * per, n)
    r1 = min(r0 + per, n)
    return r0, r1, per


def shard_rows_weighted(weights, world):
    """Contiguous row blocks of about equal total weight: list of (r0, r1) for ranks 0..world-1.

    Cut k is placed where the running weight is closest to k/world of the total, so that no rank's
    block exceeds the ideal share by more than one row's weight.  Rows of zero total weight are
    split by count."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if n == 0:
        return [(0, 0)] * world
    tot = float(w.sum())
    if tot <= 0.0:
        return [shard_rows(n, world, r)[:2] for r in range(world)]
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, world):
        target = tot * k / world
        j = int(np.searchsorted(cum, target))            # cum[j-1] < target <= cum[j]
        if j > 0 and target - cum[j - 1] < cum[j] - target:
            j -= 1
        cuts.append(min(max(j, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def lz4_row_weights(lengths):
    """Work of row i of the lz4 matrix: sum_j (tail(x_i) + len(y_j)), tail = what is left of x_i behind its
    last whole 64 KiB block (the blocks before come from the prefix snapshot)."""
    ln = np.asarray(lengths, dtype=np.float64)
    tail = np.where(ln > 65536, ln % 65536, ln)
    return len(ln) * tail + ln.sum()


def deflate_row_weights(lengths):
    """Work of row i of the gzip / zlib matrix: the pair job re-parses a seam of fixed size and re-prices the
    blocks of y_j, so a row costs about sum_j len(y_j) whatever x_i is."""
    ln = np.asarray(lengths, dtype=np.float64)
    return np.full(len(ln), ln.sum() + 1.0)


def _die(what, exc):
    """A failed rendezvous or collective must end the run with a non-zero status and a message that says
    so -- never a silent single-rank result."""
    print(f"snacc_amd: {what} failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
    raise SystemExit(3)


def init_process_group(backend, device=None):
    """torch.distributed rendezvous from the torchrun environment (127.0.0.1 when MASTER_ADDR is unset).
    Nothing here touches the GPU before the group exists.  Exits with status 3 and a message on failure."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if dist.is_initialized():
        return
    try:
        if backend == "nccl" and device is not None:
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    except Exception as e:          # noqa: BLE001
        _die(f"init_process_group({backend})", e)


def gather_tile(tile, world, group=None, out=None, async_op=False, force_collective=False):
    """All-gather one (rows, n) int32 tile per rank into a (world * rows, n) tensor on the tile's device
    (backend "nccl" = RCCL for CUDA tensors, gloo for CPU tensors).  Returns (gathered, work-or-None).
    A single rank copies; `force_collective` makes it call the collective all the same (the one-GPU RCCL
    smoke test)."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.zeros((world * tile.shape[0], tile.shape[1]), dtype=tile.dtype, device=tile.device)
    if world == 1 and not force_collective:
        out.copy_(tile)
        return out, None
    try:
        if tile.device.type == "cpu":               # gloo has no all_gather_into_tensor for every build: list form
            parts = list(out.view(world, tile.shape[0], tile.shape[1]).unbind(0))
            work = dist.all_gather(parts, tile, group=group, async_op=async_op)
        else:
            work = dist.all_gather_into_tensor(out, tile, group=group, async_op=async_op)
    except Exception as e:          # noqa: BLE001
        _die("all_gather", e)
    return out, (work if async_op else None)


def allgather_check(gathered, tile, rank, world, group=None):
    """Every rank's tile sits in its slot of the gathered tensor, on every rank (used by bench.py after its
    timed region, and by the CPU tests)."""
    import torch
    import torch.distributed as dist
    rows = tile.shape[0]
    ok = bool(torch.equal(gathered[rank * rows:(rank + 1) * rows], tile))
    if world > 1:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=tile.device)
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        except Exception as e:      # noqa: BLE001
            _die("all_reduce", e)
        ok = bool(flag.item())
    return ok


def all_pairs_sharded(n, rows_fn, group=None, device=None, weights=None, tile_rows=None, singles_fn=None):
    """Assemble the full (n, n) uint32 size matrix on every rank.

    rows_fn(r0, r1) -> either a numpy uint32 array of shape (r1-r0, n) or a torch int32/uint32 tensor
    already on `device` (the HIP backend writes straight into such a tensor).  `weights`: per-row work
    (:func:`shard_rows_weighted`); None = equal row counts.  `tile_rows`: rows per launch/gather; the
    gather of one tile is asynchronous and overlaps the next tile's rows_fn.

    `singles_fn(r0, r1)` -> numpy uint32 single sizes of the sequences [r0, r1): called ONCE, for the rank's
    own block of rows, before its tiles (on the HIP backend this is where phase A of those rows -- and of
    no others -- runs: SURVEY.md 8e, "computed once on the owning GPU").  The sizes ride on the tile gathers:
    every tile carries ceil(tile_rows / n) extra rows that hold the single sizes of its rows.  Returns
    (full, singles) then, with `singles` the (n,) uint32 array assembled from every rank's share.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if weights is None:
        blocks = [shard_rows(n, world, r)[:2] for r in range(world)]
    else:
        blocks = shard_rows_weighted(weights, world)
    r0, r1 = blocks[rank]
    per = max([b - a for a, b in blocks] + [0])          # every rank gathers `per` rows (padded)
    tile_rows = max(1, min(per, tile_rows or per))
    n_tiles = (per + tile_rows - 1) // tile_rows if per else 0
    extra = (tile_rows + n - 1) // n if (singles_fn is not None and n) else 0      # rows that carry the tile's single sizes
    own_singles = None
    if singles_fn is not None:
        own_singles = np.ascontiguousarray(singles_fn(r0, r1), dtype=np.uint32) if r1 > r0 else np.zeros(0, np.uint32)
    gathered, works, tiles = [], [], []          # tiles: kept alive until their gathers have completed
    for k in range(n_tiles):
        t0 = min(r0 + k * tile_rows, r1)
        t1 = min(t0 + tile_rows, r1)
        mine = rows_fn(t0, t1) if t1 > t0 else None
        if isinstance(mine, np.ndarray):
            mine = torch.from_numpy(np.ascontiguousarray(mine.astype(np.uint32)).view(np.int32))
            if device is not None:
                mine = mine.to(device)
        dev = mine.device if mine is not None else (device if device is not None else torch.device("cpu"))
        tile = torch.zeros((tile_rows + extra, n), dtype=torch.int32, device=dev)
        if mine is not None:
            tile[: t1 - t0] = mine.reshape(t1 - t0, n)
            if extra:
                sv = torch.from_numpy(own_singles[t0 - r0: t1 - r0].view(np.int32))
                tile.view(-1)[tile_rows * n: tile_rows * n + (t1 - t0)] = sv.to(dev)
        tiles.append(tile)
        g, work = gather_tile(tile, world, group=group, async_op=world > 1)
        gathered.append(g)
        works.append(work)
    for work in works:
        if work is not None:
            try:
                work.wait()
            except Exception as e:  # noqa: BLE001
                _die("all_gather (wait)", e)
    full = np.zeros((n, n), dtype=np.uint32)
    singles = np.zeros(n, dtype=np.uint32)
    for k, g in enumerate(gathered):
        g = g.cpu().numpy().view(np.uint32).reshape(world, tile_rows + extra, n)
        for r, (a, b) in enumerate(blocks):
            t0 = min(a + k * tile_rows, b)
            t1 = min(t0 + tile_rows, b)
            if t1 > t0:
                full[t0:t1] = g[r, : t1 - t0]
                if extra:
                    singles[t0:t1] = g[r, tile_rows:].reshape(-1)[: t1 - t0]
    return (full, singles) if singles_fn is not None else full


def _torch_device(ctx):
    import torch
    return getattr(ctx, "torch_device", None) or torch.device("cuda", ctx.device)


def all_pairs_hip(ctx, n, group=None, lengths=None, with_singles=False):
    """Sharded phase B on the HIP backend: each rank launches its row tiles on the context's own stream,
    writing into a device tensor; each finished tile is all-gathered (backend "nccl" = RCCL) while the
    next one runs.  `with_singles`: phase A too -- every rank computes the single sizes (and prefix snapshots)
    of ITS rows only (`ctx.singles_rows`; the context was created with ``defer_singles=1``) and the sizes ride on
    the tile gathers; returns (pairs, singles)."""
    import torch

    dev = _torch_device(ctx)

    def rows_fn(r0, r1):
        out = torch.zeros((max(r1 - r0, 0), n), dtype=torch.int32, device=dev)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()  # the fill must land before another stream writes
        ctx.pairs_device(r0, r1, out.data_ptr(), None)    # the context's own stream
        ctx.sync(None)                                    # complete before the collective is enqueued
        return out

    weights = lz4_row_weights(lengths) if lengths is not None else None
    tile = max(1, (4 << 20) // max(n, 1))
    return all_pairs_sharded(n, rows_fn, group=group, device=dev, weights=weights, tile_rows=tile,
                             singles_fn=ctx.singles_rows if with_singles else None)


def all_pairs_deflate_hip(ctx, n, algorithm, group=None, lengths=None):
    """Sharded phase B of the gzip / zlib path: as :func:`all_pairs_hip`, sizes include the wrapper bytes."""
    import torch
    from .hip_backend import DEFLATE

    dev = _torch_device(ctx)

    def rows_fn(r0, r1):
        out = torch.zeros((max(r1 - r0, 0), n), dtype=torch.int32, device=dev)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        ctx.deflate_pairs_device(algorithm, r0, r1, out.data_ptr(), None)
        ctx.sync(None)
        return out

    weights = deflate_row_weights(lengths) if lengths is not None else None
    tile = max(1, (1 << 20) // max(n, 1))
    return all_pairs_sharded(n, rows_fn, group=group, device=dev, weights=weights, tile_rows=tile) \
        + np.uint32(DEFLATE[algorithm][1])
