"""Dev aid (GPU box): far chains on a small set, every size against the oracle.  Usage: gpu_far_dbg.py [opt=val ...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd.hip_backend import HipContext
opts = dict(far_lanes=2, far_waves=1, far_min=0, fast_lanes=1, fast_waves=1, far_stop_pct=0)
opts.update({k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])})
L = opts.pop("L", 150000); N = opts.pop("N", 6)
seqs = [oracle.lcg_genome(1 + i, L + 1000 * i) for i in range(N)]
exp = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
with HipContext(0, **opts) as ctx:
    ctx.upload(seqs)
    p = ctx.pairs()
bad = np.argwhere(p != exp)
print("opts", opts, "mismatches", len(bad), "of", N * N)
for i, j in bad[:10]:
    print("  pair", i, j, "got", int(p[i, j]), "want", int(exp[i, j]), "diff", int(p[i, j]) - int(exp[i, j]))
