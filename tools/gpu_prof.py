"""Dev aid: one upload + one pairs() launch, for rocprofv3.  Usage: _gpu_prof.py N L ROWS LANES WAVES"""
import sys
sys.path.insert(0, '.')
import numpy as np
from bench import lcg_genomes_torch
import torch
from snacc_amd.hip_backend import HipContext
N, L, R, lanes, waves = map(int, sys.argv[1:6])
seqs = lcg_genomes_torch(N, L, 1, torch.device('cuda', 0))
ctx = HipContext(0, fast_lanes=lanes, fast_waves=waves)
ctx.upload(seqs)
p = ctx.pairs(0, R)
print("ms", ctx.last_pairs_ms(), "pairs/s", R * N / ctx.last_pairs_ms() * 1e3, "checksum", int(p.astype(np.uint64).sum()))
