"""Dev aid: one upload + one pairs() launch, for rocprofv3.
Usage: [DATA=lcg|markov|related|softmask5] [OPTS=key=value,key=value] gpu_prof.py N L ROWS LANES WAVES"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
import torch
from snacc_amd.hip_backend import HipContext
N, L, R, lanes, waves = map(int, sys.argv[1:6])
data = os.environ.get("DATA", "lcg")
dev = torch.device('cuda', 0)
if data == "markov":
    seqs = bench.markov_genomes_torch(N, L, dev)
elif data == "related":
    seqs = bench.lcg_related_torch(N, L, dev)
elif data.startswith("softmask"):
    seqs = bench.softmask_genomes(bench.lcg_genomes_torch(N, L, 1, dev), int(data[8:]))
else:
    seqs = bench.lcg_genomes_torch(N, L, 1, dev)
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("OPTS", "").split(",") if kv}
ctx = HipContext(0, fast_lanes=lanes, fast_waves=waves, **opts)
ctx.upload(seqs)
p = ctx.pairs(0, R)
print("ms", ctx.last_pairs_ms(), "pairs/s", R * N / ctx.last_pairs_ms() * 1e3, "checksum", int(p.astype(np.uint64).sum()))
