"""Dev aid (GPU box): the 2-bit kernel with two lanes per chain (fast_spec=1, the default) against the one-lane loop, both hand-scheduled
and as their C++ statements: parity on a small ragged set (every pair against the oracle), then rates on the bench shape.
Usage: gpu_spec.py N L ROWS"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
import oracle
from oracle.loader import pairs_mt
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext
N, L, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lens = [65537, 131072, 200001, 70000, 65535 + 65536, 65548, 196608, 300000, 99999, 123457]
small = [oracle.lcg_genome(11 + i, n) for i, n in enumerate(lens)]
rep = np.tile(oracle.lcg_genome(32, 5000), 40)
small += [rep, oracle.lcg_mutant(rep, 5), oracle.lcg_mutant(small[7], 3), np.tile(oracle.lcg_genome(31, 37), 3000)]
exp = pairs_mt(small, 0, len(small), 16)
for opts in ({}, {"fast_lanes": 5, "fast_waves": 2}, {"fast_asm": 0}, {"fast_spec": 0}):
    with HipContext(0, **opts) as ctx:
        ctx.upload(small)
        p = ctx.pairs()
    bad = np.argwhere(p != exp)
    print(opts, "small set parity:", len(bad) == 0, bad[:6].tolist(), flush=True)
    if len(bad):
        sys.exit(1)
seqs = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
ref = None
for name, opts in (("one lane, hand-scheduled", {"fast_spec": 0}), ("one lane, C++ statement", {"fast_spec": 0, "fast_asm": 0}),
                   ("two lanes, C++ statement", {"fast_asm": 0}), ("two lanes, hand-scheduled (default)", {})):
    with HipContext(0, **opts) as ctx:
        ctx.upload(seqs)
        ctx.pairs(0, 2)
        best = 1e9
        for _ in range(3):
            p = ctx.pairs(0, R)
            best = min(best, ctx.last_pairs_ms())
    same = True if ref is None else bool(np.array_equal(p, ref))
    ref = p if ref is None else ref
    print(f"{name:34s} ms={best:.1f} pair-compr/s={R * N / best * 1e3:.0f} equal={same}", flush=True)
