"""Dev aid (GPU box): rate of the byte kernels on the bench shape (N x L, ROWS rows) -- compact table (force_generic on ACGT),
full table (bytes_compact=0), and a 20-letter protein set -- with an oracle spot check.
Usage: gpu_bytes.py N L ROWS [option=value ...]   (e.g. bytes_gt=8 bytes_gt_wgs=2: the tables in global memory)"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
import oracle
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext
N, L, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
EXTRA = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:]}
dna = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
rng = np.random.default_rng(5)
prot = [aa[rng.integers(0, 20, L)] for _ in range(N)]
for name, seqs, opts in (("compact (ACGT, force_generic)", dna, dict(force_generic=1)),
                         ("full table (ACGT, force_generic, bytes_compact=0)", dna, dict(force_generic=1, bytes_compact=0)),
                         ("protein, 20 letters (full table)", prot, {})):
    ctx = HipContext(0, **opts, **EXTRA)
    ctx.upload(seqs)
    ctx.pairs(0, 2)
    best = 1e9
    for _ in range(2):
        p = ctx.pairs(0, R)
        best = min(best, ctx.last_pairs_ms())
    ok = all(int(p[i, j]) == oracle.lz4f_size_pair(seqs[i], seqs[j]) for i in (0, R - 1) for j in (0, 1, N // 2, N - 1))
    print(f"{EXTRA} {name:52s} hashes={ctx.num_compact_hashes} ms={best:.1f} pair-compr/s={R * N / best * 1e3:.0f} parity={ok}", flush=True)
    ctx.close()
