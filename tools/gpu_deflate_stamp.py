"""Dev aid: where a gzip / zlib pair job spends its time (needs `make -C snacc_amd/csrc stamp`).
Usage: gpu_deflate_stamp.py N L"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np
import oracle
from snacc_amd import hip_backend as hb
hb.LIB_PATH = hb.LIB_PATH.replace("libsnacc_hip.so", "libsnacc_hip_stamp.so")
N, L = int(sys.argv[1]), int(sys.argv[2])
seqs = [oracle.lcg_genome(1 + i, L) for i in range(N)]
Lb = hb.load()
buf = (ctypes.c_ulonglong * (64 * 8))()
with hb.HipContext(0) as ctx:
    ctx.upload(seqs)
    for alg in ("gzip", "zlib"):
        ctx.deflate_singles(alg)
        Lb.snk_debug_dfl_stamps(buf)
        a = np.array(buf[:], dtype=np.float64).reshape(64, 8)[:min(N, 64)]
        print(alg, "STANDALONE cycles(100MHz ticks): parse %.0f search %.0f sync %.0f flush_in_parse %.0f stream %.0f flush_in_stream %.0f iters %.0f total %.0f" % tuple(a.mean(0)))
        ctx.deflate_pairs(alg)
        Lb.snk_debug_dfl_stamps(buf)
        a = np.array(buf[:], dtype=np.float64).reshape(64, 8)
        print(alg, "PAIR       cycles(100MHz ticks): parse %.0f search %.0f sync %.0f flush_in_parse %.0f stream %.0f flush_in_stream %.0f iters %.0f total %.0f" % tuple(a.mean(0)))
