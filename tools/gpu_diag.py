import sys; sys.path.insert(0,'.')
import numpy as np, oracle
from snacc_amd import hip_backend as hip
lens = [65536, 65537, 131072, 200001, 30000, 35536, 12, 65535 + 65536, 65548, 4, 196608]
seqs = [oracle.lcg_genome(11 + i, n) for i, n in enumerate(lens)]
exp = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
for opts in ({}, {"bytes_compact": 0}, {"bytes_legacy": 1}):
    with hip.HipContext(0, **opts) as ctx:
        ctx.upload(seqs); s = ctx.singles(); p = ctx.pairs()
    bad = np.argwhere(p != exp)
    print(opts, "singles ok", np.array_equal(s, [oracle.lz4f_size(x) for x in seqs]), "bad pairs:", [(int(i), int(j), lens[i] + lens[j], int(p[i, j]), int(exp[i, j])) for i, j in bad])
