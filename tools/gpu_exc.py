"""Dev aid: rate of the pair kernel on 1 Mbp genomes with non-ACGT places against pure ACGT ones.
Usage: gpu_exc.py N L ROWS [kind ...] [option=value ...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import oracle
from snacc_amd.hip_backend import HipContext
N, L, R0 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
base = [oracle.lcg_genome(1 + i, L) for i in range(N)]
def variant(kind):
    # (every kind draws its places from its own generator: the same data whatever else is on the command line -- until round 4 one
    # generator served all kinds in turn, and a set's rate moved by 10 % with the ORDER of the kinds)
    import zlib
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + 11)
    out = []
    for a in base:
        a = a.copy()
        if kind in ("n10x100", "n10x100+iupac20"):
            for s0 in rng.integers(0, L - 200, 10):
                a[s0:s0 + 100] = ord("N")
        if kind == "n1x1000":
            s0 = int(rng.integers(0, L - 2000)); a[s0:s0 + 1000] = ord("N")
        if kind in ("iupac20", "n10x100+iupac20"):
            a[rng.integers(0, L, 20)] = rng.choice(np.frombuffer(b"RYKMSWN", dtype=np.uint8), 20)
        if kind == "lower":                                # a lower-case set: the 2-bit kernel with the lower-case LUTs
            a |= 0x20
        if kind.startswith("soft"):                        # soft-masked: PCT % lower case in runs of about 500 bases
            pct = int(kind[4:])
            for s0 in rng.integers(0, L - 600, max(1, L * pct // 100 // 500)):
                ln = int(rng.integers(300, 700))
                a[s0:s0 + ln] |= 0x20
        if kind.startswith("iupac") and kind not in ("iupac20",):
            k = int(kind[5:])
            a[rng.integers(0, L, k)] = rng.choice(np.frombuffer(b"RYKMSWN", dtype=np.uint8), k)
        out.append(a)
    return out
ref = None
KINDS = [k for k in sys.argv[4:] if "=" not in k] or ["pure", "n1x1000", "n10x100", "iupac20", "n10x100+iupac20", "iupac100"]
OPTS = {k.split("=")[0]: int(k.split("=")[1]) for k in sys.argv[4:] if "=" in k}       # any context option, e.g. exc_limit=0 force_generic=1 bytes_gt=3
for kind in KINDS:
    seqs = variant(kind)
    ctx = HipContext(0, **OPTS)
    ctx.upload(seqs)
    ctx.pairs(0, 2)
    best = 1e9
    R = R0 if R0 != 84 else ctx.fast_chains()             # one row per chain of a CU: a set with other-case stretches has 83 (bench.py does the same)
    for _ in range(2):
        p = ctx.pairs(0, R)
        best = min(best, ctx.last_pairs_ms())
    ok = int(p[R - 1, N // 2]) == oracle.lz4f_size_pair(seqs[R - 1], seqs[N // 2])
    rate = R * N / best * 1e3
    ref = ref or rate
    import ctypes
    from snacc_amd import hip_backend
    L_ = hip_backend.load()
    if hasattr(L_, "snk_debug_stats"):
        st = (ctypes.c_ulonglong * 64)()
        L_.snk_debug_stats(st)
        names = ["steady exits", "general probes", "sentinel reads", "flushes", "byte matches", "site arrivals", "sentinel puts", "ovf-only puts"]
        print("   stats (upload + 4 launches):", {nm: int(st[i]) for i, nm in enumerate(names) if i != 7})
        if st[40]:
            print("   probe stamps (cycles, first lane): windows %d, owed put %d, key %d, get+put %d, finish %d" % tuple(int(st[40 + i]) for i in range(5)))
        if st[23]:
            print(f"   other-case mode: {st[23] / st[7]:.1%} of the wave cycles; {int(st[50]):,} runs, {st[48] / max(1, st[50]):.0f} wave trips per run, "
                  f"{st[49] / max(1, st[48]):.1f} lanes per trip, {st[23] / max(1, st[48]):,.0f} cycles per wave trip")
        if st[61]:
            print(f"   table swaps (other-case mode on a table in LDS): {int(st[61]):,} per first lanes; in {st[59] / st[61]:,.0f} cycles, out {st[60] / st[61]:,.0f}, "
                  f"whole mode {st[23] / st[61]:,.0f} per swap; lanes in the mode per swap: {st[49] / max(1, st[48]):.1f}")
        if st[14]:
            print(f"   wave cycles {int(st[7]):,}: in the loop {st[13] / st[7]:.1%}, loop entries {int(st[14]):,}, per entry: outside {(st[7] - st[13]) / st[14]:,.0f} cycles"
                  f" (finish {st[24] / st[14]:,.0f}, general rounds {st[25] / st[14]:,.0f} in {st[26] / st[14]:.2f} rounds)"
                  f"; per round {st[25] / max(1, st[26]):,.0f} cycles, of which inside the probe {st[28] / max(1, st[26]):,.0f}; loop prologue {st[27] / st[14]:,.0f}, top of the outer loop {st[29] / st[14]:,.0f}")
    print(f"{OPTS if OPTS else ''} {kind:18s} rows={R} packed={ctx.num_packed}/{N} ms={best:.1f} pairs/s={rate:.0f} ({rate / ref * 100:.0f}% of pure) parity={ok}", flush=True)
    ctx.close()
