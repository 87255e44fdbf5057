"""Dev aid: re-create gpu_deflate_fuzz.py seeds and show which option sets disagree with the codec.
Usage: gpu_deflate_diag.py SEED [SEED ...]"""
import sys, gzip, zlib, types
sys.path.insert(0, '.')
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from snacc_amd import hip_backend as hip
sys.path.insert(0, 'tests')
from fuzzgen import make_set as make

for seed in map(int, sys.argv[1:]):
    seqs = make(seed); raw = [bytes(s) for s in seqs]; n = len(seqs)
    print("seed", seed, [len(s) for s in seqs])
    with ThreadPoolExecutor(16) as pool:
        exp = {alg: np.array(list(pool.map(lambda ab: len(fn(ab[0] + ab[1])), [(a, b) for a in raw for b in raw])), dtype=np.uint32).reshape(n, n)
               for alg, fn in (("gzip", gzip.compress), ("zlib", zlib.compress))}
    for opts in ({}, {"deflate_serial": 1}, {"deflate_kmer": 0}, {"deflate_serial": 1, "deflate_kmer": 0}):
        for rep in range(2):
            with hip.HipContext(0, **opts) as ctx:
                ctx.upload(seqs)
                for alg in ("gzip", "zlib"):
                    p = ctx.deflate_pairs(alg)
                    bad = np.argwhere(p != exp[alg])
                    print("  ", opts, rep, alg, "bad pairs:", [(int(i), int(j), int(p[i, j]) - int(exp[alg][i, j])) for i, j in bad][:8])
