"""Dev aid (needs `make -C snacc_amd/csrc stamp`): compare the stored symbol stream of one sequence of a fuzz seed --
segmented pass vs serial pass vs the oracle's trace.  Usage: gpu_deflate_diag2.py SEED SEQ gzip|zlib"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np
from snacc_amd import hip_backend as hb
hb.LIB_PATH = hb.LIB_PATH.replace("libsnacc_hip.so", "libsnacc_hip_stamp.so")
from oracle import deflate as D
sys.path.insert(0, 'tests')
from fuzzgen import make_set
seed, g, alg = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seqs = make_set(seed)
L = hb.load()
L.snk_debug_dfl_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
res = {}
for name, opts in (("seg", {}), ("ser", {"deflate_serial": 1})):
    with hb.HipContext(0, **opts) as ctx:
        ctx.upload(seqs)
        if alg == "zlib": ctx.deflate_singles("gzip")          # same order of events as the fuzz
        ctx.deflate_singles(alg)
        cap = len(seqs[g]) + 1
        sym = np.zeros(cap, np.uint32); pos = np.zeros(cap, np.uint32); info = np.zeros(8, np.uint32)
        rc = L.snk_debug_dfl_stream(ctx._h, g, sym.ctypes.data, pos.ctypes.data, cap, info.ctypes.data)
        res[name] = (sym[:info[0]].copy(), pos[:info[0]].copy(), info.copy())
        print(name, "rc", rc, "info nsym/unsafe/rk/rpos/rkb/rbpos/bytes/nserial", info.tolist())
raw, osym, oblk = D.trace(seqs[g], level=9 if alg == "gzip" else 6)
ln = np.where(osym >> 31, ((osym >> 16) & 0x7fff) + 3, 1).astype(np.int64)
opos = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint32)
print("oracle nsym", len(osym), "bytes", raw)
for name in ("seg", "ser"):
    sym, pos, info = res[name]
    m = min(len(sym), len(osym))
    d = np.flatnonzero((sym[:m] != osym[:m]) | (pos[:m] != opos[:m]))
    print(name, "len", len(sym), "ndiff", len(d), "first diff vs oracle:", (int(d[0]), hex(int(sym[d[0]])), int(pos[d[0]]), hex(int(osym[d[0]])), int(opos[d[0]])) if len(d) else None)
    a = seqs[g]
    for i in d[:12]:
        q = int(pos[i]); got = int(sym[i]); where = np.flatnonzero(a[max(0, q - 3000):q + 3000] == (got & 0xff)) + max(0, q - 3000) - q if got < 256 else []
        print("    idx", int(i), "pos", q, "got", hex(got), "want", hex(int(osym[i])), "byte at pos", hex(int(a[q])), "nearest offsets with got-byte", list(where[np.argsort(np.abs(where))][:3]) if len(where) else None)
    print("    diff index range", (int(d[0]), int(d[-1])) if len(d) else None, "positions", (int(pos[d[0]]), int(pos[d[-1]])) if len(d) else None)
