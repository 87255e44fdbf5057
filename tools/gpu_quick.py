"""Ad-hoc GPU parity driver (development aid; the real tests are test_gpu_*.py)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd.hip_backend import HipContext

def check(name, seqs, **opts):
    ctx = HipContext(0, **opts)
    t0 = time.time(); ctx.upload(seqs); t1 = time.time()
    s = ctx.singles()
    exp_s = np.array([oracle.lz4f_size(x) for x in seqs], dtype=np.uint32)
    p = ctx.pairs(); t2 = time.time()
    exp_p = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    ok = np.array_equal(s, exp_s) and np.array_equal(p, exp_p)
    print(f"{name}: packed={ctx.num_packed}/{len(seqs)} singles_ok={np.array_equal(s, exp_s)} pairs_ok={np.array_equal(p, exp_p)} "
          f"upload={t1-t0:.3f}s pairs={t2-t1:.3f}s kernel_ms={ctx.last_pairs_ms():.2f}", flush=True)
    if not ok:
        bad = np.argwhere(p != exp_p)[:5]
        print("  singles", s[:6], exp_s[:6]); print("  bad pairs", bad.tolist(), [ (int(p[a,b]), int(exp_p[a,b])) for a,b in bad])
    ctx.close()
    return ok

rng = np.random.default_rng(7)
acgt = np.frombuffer(b'ACGT', dtype=np.uint8)
ok = True
ok &= check("tiny", [b"ACGT"*10, b"ACGTTGCA"*3, b"A", b"ACGTN"*5, b"GATTACA"*1000])
ok &= check("lcg 6x100k", [oracle.lcg_genome(1+i, 100000) for i in range(6)])
ok &= check("lcg 6x100k generic", [oracle.lcg_genome(1+i, 100000) for i in range(6)], force_generic=1)
ok &= check("ragged", [oracle.lcg_genome(11, 65536), oracle.lcg_genome(12, 65537), oracle.lcg_genome(13, 131072),
                       oracle.lcg_genome(14, 200001), oracle.lcg_genome(15, 30000), oracle.lcg_genome(16, 35536),
                       oracle.lcg_genome(17, 12), oracle.lcg_genome(18, 65535+65536)])
mixed = [oracle.lcg_genome(21, 150000), rng.integers(0, 256, 140000, dtype=np.uint8),
         np.concatenate([oracle.lcg_genome(22, 70000), np.frombuffer(b'N'*500, dtype=np.uint8), oracle.lcg_genome(23, 70000)]),
         np.tile(oracle.lcg_genome(24, 700), 300), rng.choice(np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8), 90000)]
ok &= check("mixed", mixed)
rep = [np.tile(oracle.lcg_genome(31, 37), 5000), np.tile(oracle.lcg_genome(32, 5000), 40), oracle.lcg_mutant(np.tile(oracle.lcg_genome(32, 5000), 40), 5),
       np.frombuffer(b'A'*300000, dtype=np.uint8), oracle.lcg_genome(33, 250000)]
ok &= check("repeats (long matches)", rep)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
