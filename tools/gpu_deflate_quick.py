"""Dev aid: GPU gzip/zlib sizes vs the deflate oracle on a few input sets."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import oracle
from oracle import deflate as D
from snacc_amd import hip_backend as hip

rng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
sets = {
    "tiny": [b"ACGT" * 10, b"ACGTTGCA" * 3, b"A", b"", b"ACGTN" * 5, b"GATTACA" * 100, b"ACG", b"AC"],
    "lcg": [oracle.lcg_genome(1 + i, n) for i, n in enumerate([1000, 30000, 70000, 100000, 140000, 65536, 65537, 200000])],
    "big": [oracle.lcg_genome(40 + i, n) for i, n in enumerate([1000000, 3000001, 300000])] + [np.tile(oracle.lcg_genome(50, 1000), 400), np.full(200000, 65, dtype=np.uint8)],
    "mix": [rng.integers(0, 256, 50000, dtype=np.uint8), rng.choice(ACGT, 90000), np.tile(rng.choice(ACGT, 700), 200),
            rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), 80000), np.repeat(rng.choice(ACGT, 3000), 40)[:100000].copy()],
}
OPTS = {}
args = [a for a in sys.argv[1:] if '=' not in a]
for a in sys.argv[1:]:
    if '=' in a: OPTS[a.split('=')[0]] = int(a.split('=')[1])
which = args or list(sets)
for name in which:
    seqs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else s for s in sets[name]]
    n = len(seqs)
    for alg, fn in (("gzip", D.gzip_size), ("zlib", D.zlib_size)):
        t = time.time()
        with hip.HipContext(0, **OPTS) as ctx:
            ctx.upload(seqs)
            s = ctx.deflate_singles(alg)
            t1 = time.time() - t
            p = ctx.deflate_pairs(alg)
        t2 = time.time() - t
        es = np.array([fn(x) for x in seqs], dtype=np.uint32)
        ep = np.array([[fn(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
        bs, bp = np.flatnonzero(s != es), np.argwhere(p != ep)
        print(f"{name:5s} {alg}: singles bad {len(bs)}/{n} pairs bad {len(bp)}/{n*n}  ({t1:.2f}s, {t2:.2f}s)")
        for i in bs[:4]: print("   single", i, len(seqs[i]), int(s[i]), int(es[i]))
        for i, j in bp[:6]: print("   pair", i, j, len(seqs[i]), len(seqs[j]), int(p[i, j]), int(ep[i, j]))
