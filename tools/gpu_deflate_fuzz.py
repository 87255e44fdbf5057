"""Randomised GPU stress of the gzip / zlib path against the real codec (stdlib gzip / zlib):
gpu_deflate_fuzz.py SEED0 NSEEDS.  Sequence sets as in gpu_fuzz.py, all singles and all pairs."""
import sys, time, gzip, zlib
sys.path.insert(0, '.')
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from snacc_amd import hip_backend as hip

sys.path.insert(0, 'tests')
from fuzzgen import make_set


def one(seed, pool):
    seqs = make_set(seed)
    n = len(seqs)
    raw = [bytes(s) for s in seqs]
    bad = []
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        for alg, fn in (("gzip", gzip.compress), ("zlib", zlib.compress)):
            s, p = ctx.deflate_singles(alg), ctx.deflate_pairs(alg)
            es = np.array(list(pool.map(lambda a: len(fn(a)), raw)), dtype=np.uint32)
            ep = np.array(list(pool.map(lambda ab: len(fn(ab[0] + ab[1])), [(a, b) for a in raw for b in raw])), dtype=np.uint32).reshape(n, n)
            if not (np.array_equal(s, es) and np.array_equal(p, ep)):
                bad.append((alg, np.flatnonzero(s != es)[:4].tolist(), np.argwhere(p != ep)[:4].tolist()))
    return n, [len(x) for x in seqs], bad


seed0, nseeds = int(sys.argv[1]), int(sys.argv[2])
fails = 0
t0 = time.time()
with ThreadPoolExecutor(16) as pool:
    for seed in range(seed0, seed0 + nseeds):
        n, lens, bad = one(seed, pool)
        if bad:
            fails += 1
            print(f"seed {seed} FAIL lens={lens} {bad}", flush=True)
        else:
            print(f"seed {seed} n={n} ok ({time.time() - t0:.0f}s)", flush=True)
print("ALL OK" if not fails else f"{fails} FAILED", fails)
