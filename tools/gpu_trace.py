"""Dev aid: run ONE single-sequence job on the trace build and print the recorded events.
Usage: gpu_trace.py file.npy FROM_POS"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np
from snacc_amd import hip_backend as hb
hb.LIB_PATH = hb.LIB_PATH.replace("libsnacc_hip.so", "libsnacc_hip_trace.so")
a = np.load(sys.argv[1]); frm = int(sys.argv[2])
L = hb.load()
L.snk_debug_trace.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_int]
ctx = hb.HipContext(0, fast_lanes=1, fast_waves=1)
buf = (ctypes.c_uint * (1 + 4 * 4096))()
L.snk_debug_trace(frm, buf, 0)
ctx.upload([a])
print("gpu size", int(ctx.singles()[0]))
L.snk_debug_trace(0, buf, 1)
n = min(buf[0], 4096)
# tags 5 (tight loop: wc, rb, nx offset) and 8 (general probe: r0, r1, nx) are printed raw
names = {1: "BLOCK_END op anchor payload|endcode<<28", 2: "MATCH_SLOW cur cand f<<24|e2", 3: "TIGHT cur cand f|m|valid|e2", 4: "ITER cur cand f|m|valid|e2"}
for i in range(n):
    t, x, y, z = buf[1 + 4 * i: 5 + 4 * i]
    if t >= 5:
        print(f"  raw{t} {x:#010x} {y:#010x} {z:#010x}  ({x} {y} {z})")
    elif t == 1:
        print(f"BLOCK_END op={x} anchor={y} payload={z & 0xFFFFFFF} endcode={z >> 28}")
    else:
        print(f"{names[t].split()[0]:10s} cur={x} cand={y} f={z >> 24} m={(z >> 23) & 1} valid={(z >> 22) & 1} e2={z & 0x3FFFFF}")
