"""Dev aid (GPU box): the 2-bit kernel with chains beyond the LDS (far chains: tables in global memory, extra waves).
Times `ROWS x N` pair launches for several far configurations, checks every size against the LDS-only launch and a
sample against the oracle.
Usage: tools/gpu_far.py N L ROWS cfg...      cfg = far_lanes x far_waves (0x0 = LDS waves only), optionally :dyn to
force the atomic queue on the LDS-only run."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
import torch
import oracle
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext

N, L, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfgs = sys.argv[4:] or ["0x0", "8x4"]
seqs = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
ref = None
for cfg in cfgs:
    dyn = cfg.endswith(":dyn")
    fl, fw = map(int, cfg.split(":")[0].split("x"))
    opts = {}
    if fl:
        opts.update(far_lanes=fl, far_waves=fw)
    if dyn:
        opts.update(fast_dynamic=1)
    ctx = HipContext(0, **opts)
    ctx.upload(seqs)
    ctx.pairs(0, min(R, 84))
    best, p = 1e9, None
    for rep in range(2):
        t0 = time.time()
        p = ctx.pairs(0, R)
        best = min(best, ctx.last_pairs_ms())
    if ref is None:
        ref = p
        js = [0, 1, N // 2, N - 1]
        ok = all(int(p[i, j]) == oracle.lz4f_size_pair(seqs[i], seqs[j]) for i in (0, R - 1) for j in js)
    else:
        ok = bool(np.array_equal(p, ref))
    rate = R * N / (best * 1e-3)
    print(f"far={fl}x{fw}{' dyn' if dyn else ''} extra_chains/CU={fl * fw} ms={best:.2f} pair-compr/s={rate:.0f} "
          f"frac={rate * (2 * L + 4) / 8e12:.4f} parity={ok}", flush=True)
    ctx.close()
