"""Dev aid (GPU box): why the waves leave the loop on a soft-masked set -- the stats build's service reasons (C++ statement of the
loops: same trips) and cycle parts (hand-scheduled loops), per stretch and wave.
Usage: SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_stats.so python tools/gpu_exc_account.py [N L PCT]"""
import ctypes
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd import hip_backend
from snacc_amd.hip_backend import HipContext
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
PCT = int(sys.argv[3]) if len(sys.argv) > 3 else 5
OPTS = {k.split('=')[0]: int(k.split('=')[1]) for k in sys.argv[4:] if '=' in k}       # context options, e.g. fast_spec=0
rng = np.random.default_rng(11)
seqs = []
nst = max(1, L * PCT // 100 // 500)
for i in range(N):
    a = oracle.lcg_genome(1 + i, L).copy()
    for s0 in rng.integers(0, L - 600, nst):
        a[s0:s0 + int(rng.integers(300, 700))] |= 0x20
    seqs.append(a)
lib = hip_backend.load()
if not hasattr(lib, "snk_debug_stats"):
    sys.exit("not the stats build")
def run(asm):
    st = (ctypes.c_ulonglong * 64)()
    with HipContext(0, fast_asm=asm, **OPTS) as ctx:
        ctx.upload(seqs)
        lib.snk_debug_stats(st)
        rows = ctx.fast_chains()
        ctx.pairs(0, rows)
        ms = ctx.last_pairs_ms()
        lib.snk_debug_stats(st)
    return rows, ms, [int(v) for v in st]
rows, ms, a = run(1)
_, ms0, c = run(0)
pairs = rows * N
waves = pairs / 21.0
per = nst * waves            # stretches x waves
print(f"soft{PCT}: rows {rows}, kernel {ms:.1f} ms (C++ loops {ms0:.1f} ms), {nst} stretches per sequence")
print(f"per stretch and wave: wave cycles {a[7] / per:,.0f}; loop entries {a[14] / per:.1f}; swaps {a[61] / per:.2f}; flushes {a[3] / per:.2f}; site arrivals (lanes) {c[5] / per:.1f}")
print(f"  in the loops {a[13] / per:,.0f}; finish {a[24] / per:,.0f}; rounds {a[25] / per:,.0f} ({a[26] / per:.1f} rounds, inside the general probe {a[28] / per:,.0f}); "
      f"prologue {a[27] / per:,.0f}; top {a[29] / per:,.0f}; mode {a[23] / per:,.0f} (swap in {a[59] / per:,.0f}, out {a[60] / per:,.0f})")
names = {16: "literal run >= 15", 17: "back-extension 4", 18: "output budget", 19: "12 equal bases", 20: "block end", 21: "other limit", 22: "seam straddle"}
print("  service requests per stretch and wave (lanes):", {v: round(c[k] / per, 2) for k, v in names.items()})
print("  raw:", {i: a[i] for i in range(64) if a[i]}, {i: c[i] for i in range(64) if c[i]})
