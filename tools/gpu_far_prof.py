"""Dev aid: one upload + one pairs(0, ROWS) launch with far chains, for rocprofv3.  Usage: gpu_far_prof.py N L ROWS FAR_LANES FAR_WAVES [opt=val...]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext
N, L, R, fl, fw = map(int, sys.argv[1:6])
opts = {k: int(v) for k, v in (a.split("=") for a in sys.argv[6:])}
if fl:
    opts.update(far_lanes=fl, far_waves=fw)
seqs = lcg_genomes_torch(N, L, 1, torch.device('cuda', 0))
ctx = HipContext(0, **opts)
ctx.upload(seqs)
p = ctx.pairs(0, R)
print("ms", ctx.last_pairs_ms(), "pair-compr/s", R * N / ctx.last_pairs_ms() * 1e3, "checksum", int(p.astype(np.uint64).sum()))
