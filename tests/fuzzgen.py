"""Random sequence sets for the gzip / zlib stress runs (test infrastructure; used by tools/gpu_deflate_fuzz.py,
tools/gpu_deflate_diag*.py and the regression tests that replay a seed)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def gen(rng, n, kind):
    if n == 0:
        return np.zeros(0, np.uint8)
    if kind == "acgt":
        return rng.choice(ACGT, n)
    if kind == "acgtn":
        a = rng.choice(ACGT, n)
        for _ in range(int(rng.integers(1, 6))):
            s = int(rng.integers(0, n)); a[s:s + int(rng.integers(1, 400))] = ord("N")
        return a
    if kind == "bytes":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "aa":
        return rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), n)
    if kind == "runs":
        return np.repeat(rng.choice(ACGT, n // 20 + 1), rng.integers(1, 600, n // 20 + 1))[:n].copy()
    if kind == "repeat":
        unit = rng.choice(ACGT, int(rng.integers(1, 3000)))
        a = np.tile(unit, n // len(unit) + 1)[:n].copy()
        m = rng.random(n) < rng.choice([0.0, 0.001, 0.02])
        a[m] = rng.choice(ACGT, int(m.sum()))
        return a
    if kind == "mix":
        parts, tot = [], 0
        while tot < n:
            k = str(rng.choice(["acgt", "acgtn", "bytes", "repeat", "runs", "aa"]))
            ln = int(rng.integers(1, 60000)); parts.append(gen(rng, ln, k)); tot += ln
        return np.concatenate(parts)[:n]
    raise ValueError(kind)


def rand_len(rng):
    c = rng.integers(0, 7)
    if c == 0: return int(rng.integers(0, 40))
    if c == 1: return int(rng.integers(40, 33000))
    if c == 2: return int(32768 * rng.integers(1, 5) + rng.integers(-300, 301))
    if c == 3: return int(rng.integers(64000, 67000))
    if c == 4: return int(rng.integers(500, 700))
    return int(rng.integers(65537, 200000))



def make_set(seed):
    """The sequence set of fuzz seed `seed` (deterministic)."""
    rng = np.random.default_rng(seed)
    kinds = ["acgt", "acgtn", "bytes", "aa", "repeat", "runs", "mix"]
    n = int(rng.integers(4, 9))
    seqs = [gen(rng, rand_len(rng), str(rng.choice(kinds))) for _ in range(n)]
    for _ in range(int(rng.integers(0, 3))):
        src = seqs[int(rng.integers(0, n))]
        if len(src) < 100:
            continue
        a = src.copy()
        hit = rng.random(len(a)) < rng.choice([0.0, 0.001, 0.02])
        a[hit] = rng.choice(ACGT, int(hit.sum()))
        seqs[int(rng.integers(0, n))] = a[int(rng.integers(0, len(a) // 2)):]
    return seqs
