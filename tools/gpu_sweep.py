"""Dev aid: time the pair kernel for several (lanes, waves) settings.  Usage: _gpu_sweep.py N L ROWS cfg..."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd.hip_backend import HipContext
N, L, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfgs = [tuple(map(int, c.split('x'))) for c in sys.argv[4:]] or [(9, 4)]
seqs = [oracle.lcg_genome(1 + i, L) for i in range(N)]
exp = None
for lanes, waves in cfgs:
    ctx = HipContext(0, fast_lanes=lanes, fast_waves=waves)
    ctx.upload(seqs)
    ctx.pairs(0, min(R, 2))
    best = 1e9
    for rep in range(2):
        p = ctx.pairs(0, R)
        best = min(best, ctx.last_pairs_ms())
    if exp is None:
        exp = np.array([[oracle.lz4f_size_pair(seqs[i], seqs[j]) for j in range(N)] for i in range(min(R, 2))], dtype=np.uint32)
    ok = np.array_equal(p[:exp.shape[0]], exp)
    rate = R * N / (best * 1e-3)
    print(f"lanes={lanes} waves={waves} chains/WG={lanes*waves} ms={best:.2f} pairs/s={rate:.0f} GB/s_alg={rate*2*L/1e9:.1f} parity={ok}", flush=True)
    ctx.close()
