"""ctypes loader for oracle/liboracle_deflate.so -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Sizes of the reference's gzip and zlib paths (ref:snacc/pairwise_ncd.py:73-78):
``gzip_size(x[, y]) == len(gzip.compress(x + y))`` and ``zlib_size(x[, y]) == len(zlib.compress(x + y))``
for the zlib 1.2.11 of this image."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_deflate.so")
_lib = None

GZIP_LEVEL, GZIP_WRAPPER = 9, 18      # gzip.compress: level 9, 10 B header + 8 B trailer
ZLIB_LEVEL, ZLIB_WRAPPER = 6, 6       # zlib.compress: level 6,  2 B header + 4 B adler32


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("deflate_oracle.c", "deflate_rules.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_deflate.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, u64 = ctypes.c_void_p, ctypes.c_uint64
        L.dfl_oracle_raw_size.restype = u64
        L.dfl_oracle_raw_size.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_int]
        L.dfl_oracle_trace.restype = u64
        L.dfl_oracle_trace.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_int, vp, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_size_t), vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.dfl_oracle_trace_from.restype = u64
        L.dfl_oracle_trace_from.argtypes = [vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp, ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_size_t)]
        L.dfl_rules_pair_size.restype = u64
        L.dfl_rules_pair_size.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_int]
        L.dfl_rules_check_kpass.restype = u64
        L.dfl_rules_check_kpass.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(u64)]
        L.dfl_rules_raw_size.restype = u64
        L.dfl_rules_raw_size.argtypes = [vp, ctypes.c_size_t, ctypes.c_int]
        L.dfl_oracle_bytes_behind_end.restype = None
        L.dfl_oracle_bytes_behind_end.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, vp]
        L.dfl_oracle_window_trace.restype = ctypes.c_size_t
        L.dfl_oracle_window_trace.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, vp, ctypes.c_size_t]
        L.dfl_oracle_block_bits.restype = u64
        L.dfl_oracle_block_bits.argtypes = [vp, vp]
        _lib = L
    return _lib


def _arr(x):
    if x is None:
        return None
    return np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray, memoryview)) else x,
                                dtype=np.uint8)


def raw_size(x, y=None, level=9):
    """Bytes of the raw deflate stream of x (+ y) at `level` (4..9)."""
    a, b = _arr(x), _arr(y)
    return int(lib().dfl_oracle_raw_size(a.ctypes.data, a.size, b.ctypes.data if b is not None else None,
                                         b.size if b is not None else 0, level))


def gzip_size(x, y=None):
    return raw_size(x, y, GZIP_LEVEL) + GZIP_WRAPPER


def zlib_size(x, y=None):
    return raw_size(x, y, ZLIB_LEVEL) + ZLIB_WRAPPER


def trace(x, y=None, level=9):
    """(raw size, symbols u32[], block bits u64[]); symbol = literal byte or 0x80000000 | (len-3) << 16 | dist."""
    a, b = _arr(x), _arr(y)
    n = a.size + (b.size if b is not None else 0)
    sym = np.zeros(n + 1, dtype=np.uint32)
    blk = np.zeros(n // 16383 + 2, dtype=np.uint64)
    ns, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    r = lib().dfl_oracle_trace(a.ctypes.data, a.size, b.ctypes.data if b is not None else None,
                               b.size if b is not None else 0, level, sym.ctypes.data, sym.size, ctypes.byref(ns),
                               blk.ctypes.data, blk.size, ctypes.byref(nb))
    return int(r), sym[:ns.value], blk[:nb.value]


def trace_from(x, start, level=9):
    """Symbols of a parser that starts at `start` right behind a (fictional) match, all earlier positions in the
    hash chains (a segment job of the GPU path)."""
    a = _arr(x)
    sym = np.zeros(a.size - start + 1, dtype=np.uint32)
    ns = ctypes.c_size_t(0)
    lib().dfl_oracle_trace_from(a.ctypes.data, a.size, start, level, sym.ctypes.data, sym.size, ctypes.byref(ns))
    return sym[:ns.value]


def rules_raw_size(x, level=9):
    """Raw deflate size by the RULES the GPU kernel applies (oracle/deflate_rules.c), not by zlib's data structures."""
    a = _arr(x)
    return int(lib().dfl_rules_raw_size(a.ctypes.data, a.size, level))


def rules_pair_size(x, y, level=9):
    """Raw deflate size of x+y by the PAIR job of the GPU path as a CPU program: restart from x's own stream, seam parse
    over per-sequence chains plus the seam positions, resynchronisation with y's own stream, pricing of the splice."""
    a, b = _arr(x), _arr(y)
    return int(lib().dfl_rules_pair_size(a.ctypes.data, a.size, b.ctypes.data, b.size, level))


def rules_check_kpass(x, level=9):
    """(violations, probes decided by the K-pass): where the six-byte shortcut of the gzip kernel would decide a probe,
    does it decide like the full chain walk?"""
    a = _arr(x)
    nk = ctypes.c_uint64(0)
    v = lib().dfl_rules_check_kpass(a.ctypes.data, a.size, level, ctypes.byref(nk))
    return int(v), int(nk.value)


def bytes_behind_end(x, level=9):
    """The 258 bytes zlib's window holds right behind the last input byte when the stream is finished."""
    a = _arr(x)
    out = np.zeros(258, dtype=np.uint8)
    lib().dfl_oracle_bytes_behind_end(a.ctypes.data, a.size, level, out.ctypes.data)
    return out


def window_trace(x, level=9):
    """Per block flush: (stream position of the loop top that closed it, stream position of zlib's window[0])."""
    a = _arr(x)
    win = np.zeros(a.size // 16383 + 2, dtype=np.uint64)
    k = lib().dfl_oracle_window_trace(a.ctypes.data, a.size, level, win.ctypes.data, win.size)
    return [(int(w) >> 32, int(w) & 0xffffffff) for w in win[:k]]


def block_bits(lfreq, dfreq):
    lf = np.ascontiguousarray(lfreq, dtype=np.uint16); df = np.ascontiguousarray(dfreq, dtype=np.uint16)
    assert lf.size == 286 and df.size == 30
    return int(lib().dfl_oracle_block_bits(lf.ctypes.data, df.ctypes.data))
