"""Dev aid (GPU box): ONE launch of a byte kernel on the bench shape, for rocprofv3 (--kernel-trace --stats / --pmc).
Usage: gpu_bytes_prof.py full|compact|protein N L ROWS [option=value ...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext
mode, N, L, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
extra = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[5:]}
if mode == "protein":
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    rng = np.random.default_rng(5)
    seqs, opts = [aa[rng.integers(0, 20, L)] for _ in range(N)], {}
else:
    seqs = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
    opts = dict(force_generic=1) if mode == "compact" else dict(force_generic=1, bytes_compact=0)
ctx = HipContext(0, **opts, **extra)
ctx.upload(seqs)
ctx.pairs(0, R)
print(mode, extra, "ms", ctx.last_pairs_ms(), "pair-compr/s", R * N / ctx.last_pairs_ms() * 1e3)
ctx.close()
