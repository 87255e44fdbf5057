"""Dev aid (GPU box): a set in which only SOME genomes carry exceptions (ten 100-base runs of N each): pairs of two clean genomes on the
pure kernel (split_clean=2 always, 1 = the default: when they fill the card 16 times) against every pair on the exception kernels (0).
Usage: gpu_mixed.py N L ROWS PCT_WITH_EXCEPTIONS"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd.hip_backend import HipContext
N, L, R, PCT = (int(v) for v in sys.argv[1:5])
rng = np.random.default_rng(5)
seqs = []
for i in range(N):
    a = oracle.lcg_genome(1 + i, L)
    if rng.random() * 100 < PCT:
        a = a.copy()
        for s0 in rng.integers(0, L - 200, 10):
            a[s0:s0 + 100] = ord("N")
    seqs.append(a)
for split in (2, 1, 0):
    with HipContext(0, split_clean=split) as ctx:
        ctx.upload(seqs)
        ctx.pairs(0, 2)
        best = 1e9
        for _ in range(2):
            p = ctx.pairs(0, R)
            best = min(best, ctx.last_pairs_ms())
        ok = int(p[R - 1, N // 2]) == oracle.lz4f_size_pair(seqs[R - 1], seqs[N // 2])
        print(f"split_clean={split}: {PCT} % of the genomes with N runs, rows={R} ms={best:.1f} pairs/s={R * N / best * 1e3:.0f} parity={ok}", flush=True)
