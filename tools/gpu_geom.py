"""Dev aid (GPU box): ONE round of jobs (rows = chains per workgroup, N = CUs genomes -> one job per chain) of the 2-bit pair kernel
under several workgroup geometries: how long a round takes with 1 .. 8 waves per CU (do the waves of a CU slow each other down?).
Usage: gpu_geom.py N L lanesxwaves[,key=value...] ..."""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext
N, L = int(sys.argv[1]), int(sys.argv[2])
seqs = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
ref = None
for cfg in sys.argv[3:]:
    parts = cfg.split(",")
    lanes, waves = map(int, parts[0].split("x"))
    opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in parts[1:]}
    with HipContext(0, fast_lanes=lanes, fast_waves=waves, **opts) as ctx:
        ctx.upload(seqs)
        R = lanes * waves
        ctx.pairs(0, 1)
        best = 1e9
        for _ in range(3):
            p = ctx.pairs(0, R)
            best = min(best, ctx.last_pairs_ms())
    ok = True
    if ref is None:
        ref = p
    else:
        k = min(len(ref), len(p))
        ok = bool(np.array_equal(ref[:k], p[:k]))
    print(f"{cfg:24s} chains/CU {R:3d}: round {best:7.2f} ms -> {best * 2.4e6 / 128e3:6.0f} cycles per trip (128 k trips), {R * N / best * 1e3:9.0f} pair-compr/s  same={ok}", flush=True)
