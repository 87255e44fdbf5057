"""ctypes loader for oracle/liboracle.so (test infrastructure; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


class Stats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in
                ("sequences", "search_probes", "chain_probes", "too_far", "back_ext", "bailouts")]


class Stream(ctypes.Structure):
    _fields_ = [("table", ctypes.c_uint32 * 4096), ("pos", ctypes.c_uint64), ("out", ctypes.c_uint64)]


def build(force=False):
    """Compile liboracle.so with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "lz4f_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.c_void_p
        L.snk_oracle_lz4f_size.restype = ctypes.c_uint64
        L.snk_oracle_lz4f_size.argtypes = [u8p, ctypes.c_uint64]
        L.snk_oracle_lz4f_size_stats.restype = ctypes.c_uint64
        L.snk_oracle_lz4f_size_stats.argtypes = [u8p, ctypes.c_uint64, ctypes.POINTER(Stats)]
        L.snk_oracle_lz4f_size_pair.restype = ctypes.c_uint64
        L.snk_oracle_lz4f_size_pair.argtypes = [u8p, ctypes.c_uint64, u8p, ctypes.c_uint64]
        L.snk_oracle_lcg_genome.restype = None
        L.snk_oracle_lcg_genome.argtypes = [ctypes.c_uint64, ctypes.c_uint64, u8p]
        L.snk_oracle_lcg_mutant.restype = None
        L.snk_oracle_lcg_mutant.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64, u8p]
        L.snk_oracle_stream_init.restype = None
        L.snk_oracle_stream_init.argtypes = [ctypes.POINTER(Stream)]
        L.snk_oracle_stream_run.restype = ctypes.c_int
        L.snk_oracle_stream_run.argtypes = [ctypes.POINTER(Stream), u8p, ctypes.c_uint64,
                                            ctypes.c_uint64, ctypes.POINTER(Stats)]
        L.snk_oracle_pairs_mt.restype = ctypes.c_int
        L.snk_oracle_pairs_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.snk_oracle_pairs_list_mt.restype = ctypes.c_int
        L.snk_oracle_pairs_list_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                               ctypes.c_void_p, ctypes.c_int]
        L.snk_oracle_pairs_timed.restype = ctypes.c_int
        L.snk_oracle_pairs_timed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                             ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_uint64)]
        _lib = L
    return _lib


def _buf(b):
    """bytes / bytearray / uint8 ndarray -> (address, length, keepalive)."""
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(b) if not isinstance(b, (bytes, bytearray)) else b, dtype=np.uint8)
    return a.ctypes.data if a.size else 0, int(a.size), a


def lz4f_size(b):
    p, n, keep = _buf(b)
    if n == 0:
        return 11
    return int(lib().snk_oracle_lz4f_size(p, n))


def lz4f_size_stats(b):
    p, n, keep = _buf(b)
    st = Stats()
    r = int(lib().snk_oracle_lz4f_size_stats(p, n, ctypes.byref(st)))
    return r, {k: int(getattr(st, k)) for k, _ in Stats._fields_}


def lz4f_size_pair(x, y):
    px, nx, kx = _buf(x)
    py, ny, ky = _buf(y)
    if nx + ny == 0:
        return 11
    return int(lib().snk_oracle_lz4f_size_pair(px, nx, py, ny))


def lcg_genome(seed, n):
    out = np.empty(n, dtype=np.uint8)
    if n:
        lib().snk_oracle_lcg_genome(seed, n, out.ctypes.data)
    return out


def lcg_mutant(src, seed):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.empty_like(src)
    if src.size:
        lib().snk_oracle_lcg_mutant(src.ctypes.data, seed, src.size, out.ctypes.data)
    return out


def pairs_mt(seqs, r0, r1, nthreads):
    """Frame sizes of ordered pairs (i, j), i in [r0, r1), all j -- multi-threaded (bench cpu_baseline)."""
    arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
    n = len(arrs)
    ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
    lens = (ctypes.c_uint64 * n)(*[a.size for a in arrs])
    out = np.zeros((r1 - r0, n), dtype=np.uint32)
    rc = lib().snk_oracle_pairs_mt(ptrs, lens, n, r0, r1, out.ctypes.data, nthreads)
    if rc != 0:
        raise RuntimeError("snk_oracle_pairs_mt failed")
    return out


def pairs_list_mt(seqs, ij, nthreads):
    """Frame sizes of the ordered pairs listed in ij (shape (P, 2)) -- multi-threaded."""
    arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
    n = len(arrs)
    ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
    lens = (ctypes.c_uint64 * n)(*[a.size for a in arrs])
    ij = np.ascontiguousarray(ij, dtype=np.int32).reshape(-1, 2)
    out = np.zeros(len(ij), dtype=np.uint32)
    if lib().snk_oracle_pairs_list_mt(ptrs, lens, ij.ctypes.data, len(ij), out.ctypes.data, nthreads) != 0:
        raise RuntimeError("snk_oracle_pairs_list_mt failed")
    return out


def pairs_timed(seqs, nthreads, seconds):
    """bench.py's CPU baseline: `nthreads` threads, each with its own preallocated buffer and stream state,
    compress ordered pairs of `seqs` for `seconds`.  Returns (pairs completed, wall seconds)."""
    arrs = [np.ascontiguousarray(s, dtype=np.uint8) if isinstance(s, np.ndarray)
            else np.frombuffer(bytes(s), dtype=np.uint8) for s in seqs]
    n = len(arrs)
    ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
    lens = (ctypes.c_uint64 * n)(*[int(a.size) for a in arrs])
    done, el, ck = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_uint64()
    rc = lib().snk_oracle_pairs_timed(ptrs, lens, n, int(nthreads), float(seconds), ctypes.byref(done),
                                      ctypes.byref(el), ctypes.byref(ck))
    if rc != 0:
        raise RuntimeError("snk_oracle_pairs_timed failed")
    return int(done.value), float(el.value)
