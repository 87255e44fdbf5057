"""Dev aid (GPU box): the 2-bit kernel with the wave's LDS ring of y.  Parity on a few sets against the oracle, then
the pair rate of a row tile for several (lanes x waves x ring) settings.  Usage: gpu_ring.py N L ROWS cfg...  (cfg = LxWxRING)"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd.hip_backend import HipContext

N, L, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfgs = [tuple(map(int, c.split('x'))) for c in sys.argv[4:] if c[0].isdigit()] or [(21, 4, 0), (19, 4, 4096)]
ASM = 0 if 'cxx' in sys.argv else 1
SKIP_PARITY = 'noparity' in sys.argv


def parity(name, seqs, **opts):
    with HipContext(0, **opts) as ctx:
        ctx.upload(seqs)
        p = ctx.pairs()
    exp = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    ok = np.array_equal(p, exp)
    print(f"parity {name} {opts}: {ok}", flush=True)
    if not ok:
        print("  bad", np.argwhere(p != exp)[:6].tolist())
    return ok


ok = True
small = [oracle.lcg_genome(11 + k, n) for k, n in enumerate([200001, 330000, 150000, 65537, 400003, 70000, 262144])]
rel = [oracle.lcg_genome(41, 300000)]
rel += [oracle.lcg_mutant(rel[0], 3), oracle.lcg_mutant(rel[0], 4)[1000:], np.tile(oracle.lcg_genome(32, 5000), 60)]
for ring in (() if SKIP_PARITY else (256, 1024, 4096, 8192)):
    for asm in (1, 0):
        ok &= parity("ragged", small, fast_ring=ring, fast_asm=asm, fast_lanes=3, fast_waves=2)
    ok &= parity("relatives + repeats", rel, fast_ring=ring, fast_lanes=2, fast_waves=2)
print("PARITY OK" if ok else "PARITY FAILURES", flush=True)
if not ok:
    sys.exit(1)

seqs = [oracle.lcg_genome(1 + i, L) for i in range(N)]
exp = np.array([[oracle.lz4f_size_pair(seqs[i], seqs[j]) for j in range(N)] for i in range(2)], dtype=np.uint32)
for lanes, waves, ring in cfgs:
    with HipContext(0, fast_lanes=lanes, fast_waves=waves, fast_ring=ring, fast_asm=ASM) as ctx:
        ctx.upload(seqs)
        ctx.pairs(0, 2)
        best = 1e9
        rows = R if R else ctx.fast_chains()          # ROWS = 0: as many rows as a workgroup has chains (whole rounds of jobs)
        for rep in range(2):
            p = ctx.pairs(0, rows)
            best = min(best, ctx.last_pairs_ms())
        good = np.array_equal(p[:2], exp)
        rate = rows * N / (best * 1e-3)
        print(f"lanes={lanes} waves={waves} ring={ring} rows={rows} ms={best:.2f} pairs/s={rate:.0f} parity={good}", flush=True)
        from snacc_amd import hip_backend
        import ctypes
        L_ = hip_backend.load()
        if hasattr(L_, "snk_debug_stats"):
            st = (ctypes.c_ulonglong * 32)()
            L_.snk_debug_stats(st)
            print("   stats:", {nm: int(st[i]) for i, nm in [(0, "lane exits"), (8, "ring refills"), (9, "ring restarts"), (10, "cxx lane-trips"), (11, "cxx lane-trips, wave served by ring"), (12, "cxx lane-trips in ring"),
                                                                 (16, "svc lit>=15"), (17, "svc back-ext 4"), (18, "svc budget"), (19, "svc 12 equal"), (20, "svc block end"), (21, "svc other limit"), (22, "svc straddle"),
                                                                 (7, "wave cycles"), (13, "in-loop wave cycles"), (14, "wave loop entries"), (15, "cxx wave-trips")]}, flush=True)
