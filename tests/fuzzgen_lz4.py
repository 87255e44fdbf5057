"""Random sequence sets and the kernel configurations of the lz4 GPU stress runs (test infrastructure: used by
tools/gpu_fuzz.py -- thousands of seeds per round, outside the suite -- and by the `-m gpu` test that replays two seeds
through every configuration).  `one(seed)` uploads the set of that seed under every configuration and compares all
singles and all ordered pairs with the oracle."""
import numpy as np

import oracle
from oracle.loader import pairs_mt
from snacc_amd import hip_backend as hip

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


#: every kernel family and loop form on the same set
CONFIGS = ({}, {"fast_asm": 0}, {"force_generic": 1}, {"bytes_compact": 0}, {"force_generic": 1, "bytes_legacy": 1},
           {"fast_lanes": 5, "fast_waves": 3, "cbytes_lanes": 4, "cbytes_waves": 2},
           {"force_generic": 1, "bytes_gt": 2}, {"force_generic": 1, "bytes_compact": 0, "bytes_gt": 3},      # tables in global memory
           {"force_generic": 1, "bytes_spec": 1}, {"force_generic": 1, "bytes_compact": 0, "bytes_gt": 0, "bytes_spec": 1},   # byte kernels, two lanes per chain
           {"force_generic": 1, "bytes_spec": 1, "cbytes_lanes": 3, "cbytes_waves": 2},
           {"exc_limit": 16384},                                  # dense exceptions stay on the 2-bit kernel
           {"fast_spec": 0}, {"fast_spec": 0, "fast_asm": 0},     # one lane per chain (the default has two)
           {"fast_spec": 3}, {"fast_spec": 36, "fast_lanes": 6},  # three lanes per chain (round 4; sets without exceptions, else the default)
           {"defer_singles": 1},                                  # phase A on demand
           {"split_clean": 2}, {"split_clean": 2, "fast_asm": 0}) # (round 4) pairs of two clean sequences of a set with exceptions on the pure kernel


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
    if kind == "soft":
        a = rng.choice(ACGT, n)
        for _ in range(int(rng.integers(1, 6))):
            s = int(rng.integers(0, n)); e = s + int(rng.integers(1, 3000)); a[s:e] |= 0x20
        return a
    if kind == "bytes":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "aa":
        return rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), n)
    if kind == "repeat":
        unit = rng.choice(ACGT, int(rng.integers(1, 3000)))
        a = np.tile(unit, n // len(unit) + 1)[:n].copy()
        m = rng.random(n) < rng.choice([0.0, 0.001, 0.02])
        a[m] = rng.choice(ACGT, int(m.sum()))
        return a
    if kind == "mix":
        parts, tot = [], 0
        while tot < n:
            k = str(rng.choice(["acgt", "acgtn", "bytes", "repeat", "soft"]))
            ln = int(rng.integers(1, 90000)); parts.append(gen(rng, ln, k)); tot += ln
        return np.concatenate(parts)[:n]
    raise ValueError(kind)


def rand_len(rng):
    c = rng.integers(0, 6)
    if c == 0: return int(rng.integers(0, 40))
    if c == 1: return int(rng.integers(40, 33000))
    if c == 2: return int(65536 * rng.integers(1, 4) + rng.integers(-20, 21))
    if c == 3: return int(rng.integers(60000, 70000))
    return int(rng.integers(65537, 260000))


def one(seed):
    rng = np.random.default_rng(seed)
    profile = str(rng.choice(["pure", "withN", "soft", "anything"]))
    kinds = {"pure": ["acgt", "repeat"], "withN": ["acgt", "acgtn", "repeat"], "soft": ["acgt", "soft", "acgtn"],
             "anything": ["acgt", "acgtn", "soft", "bytes", "aa", "repeat", "mix"]}[profile]
    n = int(rng.integers(6, 15))
    seqs = [gen(rng, rand_len(rng), str(rng.choice(kinds))) for _ in range(n)]
    # relatives: mutated / shifted / truncated copies of earlier members (long cross-seam matches)
    for _ in range(int(rng.integers(0, 4))):
        src = seqs[int(rng.integers(0, len(seqs)))]
        if len(src) < 100:
            continue
        a = src.copy()
        hit = rng.random(len(a)) < rng.choice([0.0, 0.0005, 0.01, 0.1])
        a[hit] = rng.choice(ACGT, int(hit.sum()))
        for _ in range(int(rng.integers(0, 4))):                    # indels
            q = int(rng.integers(0, len(a)))
            a = np.concatenate([a[:q], rng.choice(ACGT, int(rng.integers(0, 50))), a[q + int(rng.integers(0, 50)):]])
        s0 = int(rng.integers(0, min(len(a) // 2, 70000) + 1))
        seqs[int(rng.integers(0, len(seqs)))] = a[s0:]
    n = len(seqs)
    exp_s = np.array([oracle.lz4f_size(x) for x in seqs], dtype=np.uint32)
    exp_p = pairs_mt(seqs, 0, n, 16)
    bad = []
    for opts in CONFIGS:
        with hip.HipContext(0, **opts) as ctx:
            ctx.upload(seqs)
            s, p = ctx.singles(), ctx.pairs()
            info = (ctx.num_packed, ctx.num_compact_hashes)
        if not (np.array_equal(s, exp_s) and np.array_equal(p, exp_p)):
            bad.append((opts, np.argwhere(p != exp_p)[:4].tolist(), np.argwhere(s != exp_s)[:4].tolist()))
    return profile, n, [len(x) for x in seqs], info, bad


