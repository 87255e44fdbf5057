"""Row-sharded all-pairs over the GPUs of one node.

Every rank holds all sequences (1024 x 1 Mbp is 1 GB of ASCII, 256 MB packed -- nothing
against 288 GB of HBM) and computes the ordered pairs (i, j) for its contiguous block of
prefix rows i.  There is no communication inside the loop; the only exchange is ONE
all-gather of the u32 size tiles at the end (RCCL over xGMI on GPUs: 4 MB in total at
N = 1024, latency-bound).  NCD floats are computed on the host afterwards.

The same code runs on the ``gloo`` backend with CPU tensors; the CPU test-suite drives it
with a checker-provided ``rows_fn`` (world_size 2).
"""
import numpy as np


def shard_rows(n, world, rank):
    """Contiguous row block of `rank`: (r0, r1, rows_per_rank) with rows_per_rank = ceil(n / world)."""
    per = (n + world - 1) // world
    r0 = min(rank * per, n)
    r1 = min(r0 + per, n)
    return r0, r1, per


def all_pairs_sharded(n, rows_fn, group=None, device=None):
    """Assemble the full (n, n) uint32 size matrix on every rank.

    rows_fn(r0, r1) -> either a numpy uint32 array of shape (r1-r0, n) or a torch int32/uint32
    tensor already on `device` (the HIP backend writes straight into such a tensor).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    r0, r1, per = shard_rows(n, world, rank)
    mine = rows_fn(r0, r1)
    if isinstance(mine, np.ndarray):
        mine = torch.from_numpy(np.ascontiguousarray(mine.astype(np.uint32)).view(np.int32))
        if device is not None:
            mine = mine.to(device)
    mine = mine.reshape(r1 - r0, n)
    tile = torch.zeros((per, n), dtype=torch.int32, device=mine.device)
    tile[: r1 - r0] = mine
    if world == 1:
        full = tile
    else:
        full = torch.zeros((world * per, n), dtype=torch.int32, device=mine.device)
        dist.all_gather_into_tensor(full, tile, group=group)
    return full[:n].cpu().numpy().view(np.uint32)


def all_pairs_hip(ctx, n, group=None):
    """Sharded phase B on the HIP backend: each rank launches its rows on torch's current
    stream, writing into a CUDA tensor that is then all-gathered (backend "nccl" = RCCL)."""
    import torch

    dev = torch.device("cuda", ctx.device)

    def rows_fn(r0, r1):
        out = torch.zeros((max(r1 - r0, 0), n), dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()      # the fill must land before another stream writes
        # launch on the context's own stream (NULL), in row tiles of bounded job-list size, and wait:
        # the shard is complete before the collective is enqueued on torch's stream
        tile = max(1, min(r1 - r0, (4 << 20) // max(n, 1)))
        for t0 in range(r0, r1, tile):
            t1 = min(r1, t0 + tile)
            ctx.pairs_device(t0, t1, out.data_ptr() + (t0 - r0) * n * 4, None)
            ctx.sync(None)
        return out

    return all_pairs_sharded(n, rows_fn, group=group, device=dev)


def all_pairs_deflate_hip(ctx, n, algorithm, group=None):
    """Sharded phase B of the gzip / zlib path: as :func:`all_pairs_hip`, sizes include the wrapper bytes."""
    import torch
    from .hip_backend import DEFLATE

    dev = torch.device("cuda", ctx.device)

    def rows_fn(r0, r1):
        out = torch.zeros((max(r1 - r0, 0), n), dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        tile = max(1, min(r1 - r0, (1 << 20) // max(n, 1)))
        for t0 in range(r0, r1, tile):
            t1 = min(r1, t0 + tile)
            ctx.deflate_pairs_device(algorithm, t0, t1, out.data_ptr() + (t0 - r0) * n * 4, None)
            ctx.sync(None)
        return out

    return all_pairs_sharded(n, rows_fn, group=group, device=dev) + np.uint32(DEFLATE[algorithm][1])
