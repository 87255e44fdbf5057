"""Dev aid (GPU box): wall time of phase A (singles + prefix snapshots) under several launch geometries, from snk_upload_times.
With the stats build (SNACC_HIP_LIB=.../libsnacc_hip_stats.so) also the waves' cycle account of the singles launch.
Usage: gpu_singles.py N L [key=value,key=value ...]   (each further argument is one variant's option set; "-" = defaults)"""
import ctypes
import sys
sys.path.insert(0, '.')
import torch
from bench import lcg_genomes_torch
from snacc_amd import hip_backend
from snacc_amd.hip_backend import HipContext
N, L = int(sys.argv[1]), int(sys.argv[2])
variants = sys.argv[3:] or ["-"]
seqs = lcg_genomes_torch(N, L, 1, torch.device('cuda', 0))
lib = hip_backend.load()
stats = hasattr(lib, "snk_debug_stats")
want = None
for v in variants:
    opts = {} if v == "-" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",")}
    for rep in range(2):
        st = (ctypes.c_ulonglong * 64)()
        with HipContext(0, **opts) as ctx:
            if stats:
                lib.snk_debug_stats(st)
            ctx.upload(seqs)
            t = ctx.upload_times()
            if stats:
                lib.snk_debug_stats(st)
            s = ctx.singles()
        want = s if want is None else want
        ok = bool((s == want).all())
        line = f"{v:32s} rep {rep}: singles {t['singles'] * 1e3:7.1f} ms  h2d {t['h2d'] * 1e3:6.1f}  classify {t['classify'] * 1e3:5.1f}  pack {t['pack'] * 1e3:5.1f}  total {t['total'] * 1e3:7.1f}  same={ok}"
        if stats and st[31]:
            a = [int(x) for x in st]
            line += (f" | waves {a[31]}, cycles/wave {a[7] / a[31]:,.0f}, in loop {a[13] / a[7]:.1%}, entries/wave {a[14] / a[31]:.0f},"
                     f" outside per entry {(a[7] - a[13]) / max(a[14], 1):,.0f}, wave wall {a[51] / a[31] / 1e5:.1f} ms -> {a[7] / max(a[51], 1) / 10:.2f} GHz, longest wave {a[52] / 1e5:.1f} ms, first start to last end {(a[54] - ((~a[53]) & (2**64 - 1))) / 1e5:.1f} ms")
        print(line, flush=True)
