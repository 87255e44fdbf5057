"""Randomised GPU stress: random sequence sets (lengths, alphabets, repeats), ALL pairs and singles
through every kernel family and loop form (tests/fuzzgen_lz4.py), compared with the oracle.  Usage: gpu_fuzz.py SEED0 NSEEDS"""
import sys, time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from fuzzgen_lz4 import one

seed0, nseeds = int(sys.argv[1]), int(sys.argv[2])
fails = 0
t0 = time.time()
for seed in range(seed0, seed0 + nseeds):
    profile, n, lens, info, bad = one(seed)
    if bad:
        fails += 1
        print(f"seed {seed} [{profile}] n={n} lens={lens} packed/hashes={info} MISMATCH {bad}", flush=True)
    else:
        print(f"seed {seed} [{profile}] n={n} ok ({time.time() - t0:.0f}s)", flush=True)
print("FAILED" if fails else "ALL OK", fails)
sys.exit(1 if fails else 0)
