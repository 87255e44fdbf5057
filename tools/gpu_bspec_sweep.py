import sys
sys.path.insert(0, '.')
import numpy as np, torch
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext
N, L = 1024, 1000000
dna = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
for waves in (1, 2, 4):
    for lanes in (17,):
        for spec in (0, 1):
            ctx = HipContext(0, force_generic=1, bytes_gt=0, bytes_spec=spec, cbytes_waves=waves, cbytes_lanes=lanes)
            ctx.upload(dna)
            R = 21 * waves
            ctx.pairs(0, 2)
            best = 1e9
            for _ in range(2):
                p = ctx.pairs(0, R); best = min(best, ctx.last_pairs_ms())
            print(f"waves={waves} lanes={lanes} spec={spec} rows={R} ms={best:.1f} pair-compr/s={R*N/best*1e3:.0f}", flush=True)
            ctx.close()
