"""Dev aid: run the stamp (diagnostic) build and print the per-segment cycle shares of the y-only steady loop."""
import sys, ctypes
sys.path.insert(0, '.')
import torch
from bench import lcg_genomes_torch
from snacc_amd import hip_backend as hb
hb.LIB_PATH = hb.LIB_PATH.replace("libsnacc_hip.so", "libsnacc_hip_stamp.so")
seqs = lcg_genomes_torch(128, 1000000, 1, torch.device('cuda', 0))
ctx = hb.HipContext(0); ctx.upload(seqs); ctx.pairs(0, 84)
buf = (ctypes.c_ulonglong * 8)()
L = hb.load(); L.snk_debug_read_stamps.argtypes = [ctypes.c_void_p]; print("rc", L.snk_debug_read_stamps(buf))
a1, a2, a3, a4, it = [int(buf[i]) for i in range(5)]
print(f"iters {it}: top->cand {a1/it:.0f}  cand->window {a2/it:.0f}  window->LUT issued {a3/it:.0f}  bookkeeping+LUT wait {a4/it:.0f}  total {(a1+a2+a3+a4)/it:.0f} cycles/iter (stamps cost ~40 each)")
print("kernel ms", ctx.last_pairs_ms())
