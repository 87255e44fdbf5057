"""Dev aid: regenerate a fuzz seed's sequences, find the failing single, bisect the shortest failing prefix."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np, oracle
from snacc_amd import hip_backend as hip
import importlib.util
spec = importlib.util.spec_from_file_location("fz", "tools/gpu_fuzz.py")
src = open("tools/gpu_fuzz.py").read().split("seed0, nseeds =")[0]
ns = {}; exec(compile(src, "fz", "exec"), ns)
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
profile = str(rng.choice(["pure", "withN", "soft", "anything"]))
kinds = {"pure": ["acgt", "repeat"], "withN": ["acgt", "acgtn", "repeat"], "soft": ["acgt", "soft", "acgtn"],
         "anything": ["acgt", "acgtn", "soft", "bytes", "aa", "repeat", "mix"]}[profile]
n = int(rng.integers(6, 15))
seqs = []
for _ in range(n):
    ln = ns["rand_len"](rng); kind = str(rng.choice(kinds)); seqs.append((kind, ns["gen"](rng, ln, kind)))
ctx = hip.HipContext(0)
def gpu_single(a):
    ctx.upload([a]); return int(ctx.singles()[0])
for i, (kind, a) in enumerate(seqs):
    g, o = gpu_single(a), oracle.lz4f_size(a)
    if g != o:
        print(f"seq {i} kind={kind} len={len(a)} gpu={g} oracle={o}")
        lo, hi = 0, len(a)            # gpu(prefix lo) ok, gpu(prefix hi) bad
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if gpu_single(a[:mid]) == oracle.lz4f_size(a[:mid]): lo = mid
            else: hi = mid
        print("  shortest failing prefix:", hi, "block", hi // 65536, "offset in block", hi % 65536, "gpu", gpu_single(a[:hi]), "oracle", oracle.lz4f_size(a[:hi]))
        np.save(f"gpurun_out/fail_seed{seed}_seq{i}.npy", a[:hi])
        break
