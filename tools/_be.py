import ctypes, sys
sys.path.insert(0, '.')
import oracle
from snacc_amd import hip_backend
from snacc_amd.hip_backend import HipContext
N, L = 256, 1000000
seqs = [oracle.lcg_genome(1 + i, L) for i in range(N)]
lib = hip_backend.load()
st = (ctypes.c_ulonglong * 64)()
with HipContext(0) as ctx:
    ctx.upload(seqs); lib.snk_debug_stats(st)
    ctx.pairs(0, 84); print("ms", ctx.last_pairs_ms()); lib.snk_debug_stats(st)
a = [int(v) for v in st]
n = a[63]
print("block-end passes", n, "cycles top+rounds per pass", a[62] / n, "rounds", a[55] / n)
print("top %.0f | per pass: pre-iter %.0f (in %.2f rounds with an iter), iter %.0f" % (a[32] / n, a[33] / n, a[36] / n, a[34] / n))
be = 16 * 84 * N
print("block_step per block end (lane cycles): close %.0f, ageing %.0f, open %.0f; calls %d vs %d" % (a[37] / be, a[38] / be, a[39] / be, be, n))
