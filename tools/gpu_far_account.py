"""Dev aid (GPU box, stats build): cycle account of the LDS waves and of the far waves of one launch, side by side.
Usage: SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_stats.so python tools/gpu_far_account.py N L ROWS cfg...   (cfg = FAR_LANESxFAR_WAVES)"""
import ctypes
import json
import sys
sys.path.insert(0, '.')
import torch
from bench import lcg_genomes_torch
from snacc_amd import hip_backend
from snacc_amd.hip_backend import HipContext

N, L, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seqs = lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))
lib = hip_backend.load()
if not hasattr(lib, "snk_debug_stats"):
    sys.exit("not the stats build: set SNACC_HIP_LIB to libsnacc_hip_stats.so")
PROBES = 212_000           # probes of one 1 Mbp pair job (oracle statistics, bench.py's probes_per_pair)
for cfg in sys.argv[4:]:
    fl, fw = map(int, cfg.split("x"))
    opts = dict(far_lanes=fl, far_waves=fw) if fl else dict(fast_dynamic=1)
    st = (ctypes.c_ulonglong * 64)()
    with HipContext(0, **opts) as ctx:
        ctx.upload(seqs)
        ctx.pairs(0, 8)
        lib.snk_debug_stats(st)
        ctx.pairs(0, R)
        ms = ctx.last_pairs_ms()
        lib.snk_debug_stats(st)
    a = [int(v) for v in st]
    out = {"cfg": cfg, "kernel_ms": ms, "pair_compr_per_s": R * N / ms * 1e3}
    for name, o, lanes in (("lds", 0, 21), ("far", 32, fl)):
        waves, jobs = a[o + 31], a[o + 30]
        if not waves or not jobs:
            continue
        trips = jobs / lanes * PROBES * (L / 1e6)               # wave-trips, about
        out[name] = {"waves": waves, "jobs": jobs, "jobs_per_chain": jobs / (waves * lanes),
                     "wave_cycles_mean": a[o + 7] / waves, "cycles_per_trip_in_loop_est": a[o + 13] / trips,
                     "share_outside_loop": (a[o + 7] - a[o + 13]) / a[o + 7], "loop_entries": a[o + 14],
                     "cycles_per_exit": (a[o + 7] - a[o + 13]) / max(a[o + 14], 1)}
    print(json.dumps(out), flush=True)
