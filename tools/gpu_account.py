"""Dev aid (GPU box): cycle account of the 2-bit kernel's waves from the stats build (`make -C snacc_amd/csrc stats`).
Runs a row tile once with the hand-scheduled loop (cycles) and once with its C++ statement (trips and service reasons: the two
loops make the same trips), and prints cycles per trip, the cost of leaving and re-entering the loop, and why lanes ask for service.
Usage: [DATA=lcg|markov|related] SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_stats.so python tools/gpu_account.py [N L COMMIT]   (default 256 x 1 Mbp, rows = chains)"""
import ctypes
import json
import sys
sys.path.insert(0, '.')
import oracle
from snacc_amd import hip_backend
from snacc_amd.hip_backend import HipContext

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
import os
DATA = os.environ.get("DATA", "lcg")                     # lcg | markov | related: bench.py's data sets
if DATA == "lcg":
    seqs = [oracle.lcg_genome(1 + i, L) for i in range(N)]
else:
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    seqs = bench.markov_genomes_torch(N, L, "cuda:0") if DATA == "markov" else bench.lcg_related_torch(N, L, "cuda:0")
lib = hip_backend.load()
if not hasattr(lib, "snk_debug_stats"):
    sys.exit("not the stats build: set SNACC_HIP_LIB to libsnacc_hip_stats.so")


def run(asm):
    st = (ctypes.c_ulonglong * 64)()
    with HipContext(0, fast_asm=asm) as ctx:
        ctx.upload(seqs)
        lib.snk_debug_stats(st)                      # (reads and clears: drop the upload's single-sequence pass)
        rows = ctx.fast_chains()
        ctx.pairs(0, rows)
        ms = ctx.last_pairs_ms()
        lib.snk_debug_stats(st)
    return rows, ms, [int(v) for v in st]


rows, ms, a = run(1)
_, _, c = run(0)
pairs = rows * N
trips, entries = c[15], a[14]
out = {"data": DATA, "genomes": N, "length": L, "rows": rows, "pairs": pairs, "kernel_ms": ms,
       "wave_trips": trips, "loop_entries": entries, "trips_per_entry": trips / entries,
       "cycles_per_trip_in_loop": a[13] / trips,
       # two lanes per chain (the default): a chain-trip is role 0's probe plus role 1's when it counted
       "chain_trips": c[56], "second_lane_probes_counted": c[57], "probes_per_chain_trip": (c[56] + c[57]) / c[56] if c[56] else 1.0,
       "cycles_per_exit_outside_loop": (a[7] - a[13]) / entries,
       "share_outside_loop": (a[7] - a[13]) / a[7],
       # passes between two loop entries that serve a block end (a lane past its block's last probe position): top + rounds
       "block_end_passes": {"per_wave_job_of_21_pairs": a[63] * 21.0 / pairs, "cycles_top_and_rounds": a[62] / max(1, a[63]), "rounds": a[55] / max(1, a[63]),
                            "other_passes_cycles_top_and_rounds": (a[25] + a[29] - a[62]) / max(1, entries - a[63])},
       "per_exit": {"finish": a[24] / entries, "general_rounds_and_reseat": a[25] / entries, "rounds": a[26] / entries, "prologue": a[27] / entries, "top_of_the_wave_loop": a[29] / entries,
                    "elsewhere": (a[7] - a[13] - a[24] - a[25] - a[27] - a[29]) / entries},
       "service_requests_per_pair": {k: c[i] / pairs for i, k in [(16, "literal run >= 15"), (17, "back-extension 4"), (18, "output budget"),
                                                                  (19, "12 equal bases"), (20, "block end"), (21, "other limit"), (22, "seam straddle")]}}
out["collected_at_commit"] = sys.argv[3] if len(sys.argv) > 3 else "?"
print(json.dumps(out, indent=1))
