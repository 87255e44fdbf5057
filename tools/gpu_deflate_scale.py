"""Dev aid: time the GPU gzip/zlib path on N LCG genomes of L bases and check a sample against the oracle.
Usage: gpu_deflate_scale.py N L [NSAMPLE]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import oracle
from oracle import deflate as D
from snacc_amd import hip_backend as hip

N, L = int(sys.argv[1]), int(sys.argv[2])
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
seqs = [oracle.lcg_genome(1 + i, L) for i in range(N)]
rng = np.random.default_rng(0)
with hip.HipContext(0) as ctx:
    ctx.upload(seqs)
    for alg, fn in (("gzip", D.gzip_size), ("zlib", D.zlib_size)):
        t = time.time(); s = ctx.deflate_singles(alg); t1 = time.time() - t
        t = time.time(); p = ctx.deflate_pairs(alg); t2 = time.time() - t
        print(f"{alg}: N={N} L={L}  prepare+singles {t1:.2f}s   pairs {N*N} in {t2:.2f}s = {N*N/t2:.0f} pair-compr/s", flush=True)
        bad = 0
        for _ in range(NS):
            i, j = int(rng.integers(0, N)), int(rng.integers(0, N))
            e = fn(seqs[i], seqs[j]); bad += int(p[i, j]) != e
        es = fn(seqs[0]); bad += int(s[0]) != es
        print(f"   sample of {NS} pairs + 1 single vs oracle: {bad} mismatches; symmetric? {np.array_equal(p, p.T)}", flush=True)
