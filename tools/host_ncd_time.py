"""Dev aid (host only): where the NCD assembly's time goes at N = 1024."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from snacc_amd.matrix import ncd_matrix
rng = np.random.default_rng(1)
N = 1024
s32 = rng.integers(500000, 600000, N).astype(np.uint32)
p32 = rng.integers(1000000, 1200000, (N, N)).astype(np.uint32)
def t(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3
print("astype+33        %.1f ms" % t(lambda: p32.astype(np.int64) + 33))
p = p32.astype(np.int64) + 33; s = s32.astype(np.int64) + 33
print("ncd_matrix       %.1f ms" % t(lambda: ncd_matrix(s, p)))
print("min(p, p.T) i64  %.1f ms" % t(lambda: np.minimum(p, p.T)))
print("min(p, p.T) u32  %.1f ms" % t(lambda: np.minimum(p32, p32.T)))
print("contig p.T u32   %.1f ms" % t(lambda: np.ascontiguousarray(p32.T)))
print("astype f64       %.1f ms" % t(lambda: p.astype(np.float64)))
print("divide           %.1f ms" % t(lambda: p.astype(np.float64) / 3.0))
