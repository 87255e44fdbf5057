"""Dev aid: one pair of very long genomes (positions beyond 2^23) through the gzip / zlib path, against the codec.
Usage: gpu_deflate_big.py L"""
import sys, time, gzip, zlib
sys.path.insert(0, '.')
import numpy as np
import oracle
from snacc_amd import hip_backend as hip
L = int(sys.argv[1])
seqs = [oracle.lcg_genome(1, L), oracle.lcg_genome(2, L // 2 + 12345)]
raw = [bytes(s) for s in seqs]
with hip.HipContext(0) as ctx:
    ctx.upload(seqs)
    for alg, fn in (("gzip", gzip.compress), ("zlib", zlib.compress)):
        t = time.time(); s = ctx.deflate_singles(alg); p = ctx.deflate_pairs(alg); dt = time.time() - t
        es = [len(fn(r)) for r in raw]
        ep = [[len(fn(a + b)) for b in raw] for a in raw]
        print(alg, "gpu", f"{dt:.2f}s", "singles ok" if [int(v) for v in s] == es else ("singles BAD", s, es),
              "pairs ok" if p.tolist() == ep else ("pairs BAD", p.tolist(), ep), flush=True)
