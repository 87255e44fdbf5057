"""Dev aid (GPU box, stats build): how often the second lane of the byte kernels' two-lane loop counts.
Usage: SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_stats.so python tools/gpu_bspec_count.py"""
import sys, ctypes
sys.path.insert(0, '.')
import torch
from bench import lcg_genomes_torch
from snacc_amd import hip_backend
from snacc_amd.hip_backend import HipContext
dna = lcg_genomes_torch(64, 1000000, 1, torch.device("cuda", 0))
ctx = HipContext(0, force_generic=1, bytes_gt=0, bytes_spec=1)
ctx.upload(dna)
L_ = hip_backend.load()
st = (ctypes.c_ulonglong * 64)()
L_.snk_debug_stats(st)
b = [int(st[i]) for i in (58, 59, 60, 61)]
ctx.pairs(0, 8)
L_.snk_debug_stats(st)
a = [int(st[i]) - b[k] for k, i in enumerate((58, 59, 60, 61))]
print("chain trips %d, second lane counted %d (%.1f %%), role 0 offered %d (%.1f %%), role 1 offered %d (%.1f %%)" %
      (a[0], a[1], 100.0 * a[1] / max(a[0], 1), a[2], 100.0 * a[2] / max(a[0], 1), a[3], 100.0 * a[3] / max(a[0], 1)))
