"""Dev aid: throughput of the pair kernel on ragged genome lengths (0.5 .. 1.5 x L) against uniform ones,
as parsed bases per second (a pair costs about tail(x) + len(y)), for the static and the atomic-queue schedule.
Usage: gpu_ragged.py N L ROWS"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd.hip_backend import HipContext
N, L, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(5)
big = oracle.lcg_genome(7, 2 * L + N)
def run(lens, dyn, tag):
    seqs = [big[i:i + n] for i, n in enumerate(lens)]          # shifted views: same statistics, different content per row
    ctx = HipContext(0, fast_dynamic=dyn)
    ctx.upload(seqs)
    ctx.pairs(0, 2)
    best = 1e9
    for _ in range(2):
        p = ctx.pairs(0, R)
        best = min(best, ctx.last_pairs_ms())
    work = sum((lens[i] % 65536) + lens[j] for i in range(R) for j in range(N))
    i, j = R - 1, N // 2
    ok = int(p[i, j]) == oracle.lz4f_size_pair(seqs[i], seqs[j])
    print(f"{tag}: ms={best:.1f} pairs/s={R*N/best*1e3:.0f} parsed Gbases/s={work/best/1e6:.2f} parity={ok}", flush=True)
    ctx.close()
uni = [L] * N
rag = [int(v) for v in rng.integers(L // 2, 3 * L // 2, N)]
run(uni, 0, "uniform static ")
run(uni, 1, "uniform dynamic")
run(rag, 0, "ragged  static ")
run(rag, 1, "ragged  dynamic")
run(rag, -1, "ragged  auto   ")
R = N
run(uni, -1, "uniform auto, all rows")
run(rag, 0, "ragged static, all rows")
run(rag, -1, "ragged auto, all rows")
