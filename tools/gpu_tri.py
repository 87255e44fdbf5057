"""Dev aid (GPU box): three lanes per chain (fast_spec=3, C++ statement, 20 chains per wave) against two lanes per chain as the C++
statement at the same 20 chains per wave, and the shipped hand-scheduled two-lane loop at 20 and 21: parity on a small ragged set
(every pair against the oracle), then one round of jobs on the bench shape (rows = chains per CU).  With the stats build
(SNACC_HIP_LIB=.../libsnacc_hip_stats.so) also the probes per chain-trip of each C++ loop.
Usage: gpu_tri.py N L"""
import ctypes
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
import oracle
from oracle.loader import pairs_mt
from bench import lcg_genomes_torch
from snacc_amd import hip_backend
from snacc_amd.hip_backend import HipContext
N, L = int(sys.argv[1]), int(sys.argv[2])
lens = [65537, 131072, 200001, 70000, 65535 + 65536, 65548, 196608, 300000, 99999, 123457]
small = [oracle.lcg_genome(11 + i, n) for i, n in enumerate(lens)]
rep = np.tile(oracle.lcg_genome(32, 5000), 40)
small += [rep, oracle.lcg_mutant(rep, 5), oracle.lcg_mutant(small[7], 3), np.tile(oracle.lcg_genome(31, 37), 3000)]
exp = pairs_mt(small, 0, len(small), 16)
for opts in ({"fast_spec": 3}, {"fast_spec": 3, "fast_lanes": 7, "fast_waves": 2}, {"fast_spec": 3, "fast_lanes": 5, "fast_waves": 3},
             {"fast_spec": 36}, {"fast_spec": 36, "fast_lanes": 7, "fast_waves": 2}):
    with HipContext(0, **opts) as ctx:
        ctx.upload(small)
        p = ctx.pairs()
    bad = np.argwhere(p != exp)
    print(opts, "small set parity:", len(bad) == 0, bad[:6].tolist(), flush=True)
    if len(bad):
        sys.exit(1)
lib = hip_backend.load()
stats = hasattr(lib, "snk_debug_stats")
seqs = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
ref = None
for name, opts in (("two lanes, hand-scheduled, 21 x 4", {}), ("two lanes, hand-scheduled, 20 x 4", {"fast_lanes": 20}),
                   ("two lanes, C++ statement, 20 x 4", {"fast_asm": 0, "fast_lanes": 20}),
                   ("three lanes (+5, +10), C++, 20 x 4", {"fast_spec": 3, "fast_lanes": 20}),
                   ("three lanes (+5, +6), C++, 20 x 4", {"fast_spec": 36, "fast_lanes": 20}),
                   ("one lane, C++ statement, 20 x 4", {"fast_asm": 0, "fast_spec": 0, "fast_lanes": 20})):
    with HipContext(0, **opts) as ctx:
        ctx.upload(seqs)
        R = ctx.fast_chains()
        ctx.pairs(0, 2)
        st = (ctypes.c_ulonglong * 64)()
        best = 1e9
        for _ in range(3):
            if stats:
                lib.snk_debug_stats(st)
            p = ctx.pairs(0, R)
            best = min(best, ctx.last_pairs_ms())
        if stats:
            lib.snk_debug_stats(st)
    k = min(len(p), 80)
    same = True if ref is None else bool(np.array_equal(p[:k], ref[:k]))
    ref = p if ref is None else ref
    line = f"{name:36s} rows {R}: ms={best:.1f} pair-compr/s={R * N / best * 1e3:.0f} equal={same}"
    if stats and st[56]:
        a = [int(v) for v in st]
        line += (f" | chain-trips {a[56]:,}, role 1 counted {a[57] / a[56]:.3f}, role 2 counted {a[58] / a[56]:.3f} -> {(a[56] + a[57] + a[58]) / a[56]:.3f} probes per chain-trip;"
                 f" wave trips {a[15]:,}, in loop {a[13] / max(a[7], 1):.1%}, cycles per wave trip {a[13] / max(a[15], 1):.0f}, loop entries {a[14]:,}")
    print(line, flush=True)
