"""Optional ctypes binding to the liblz4 *binary* shipped in the image (1.9.3).

TEST INFRASTRUCTURE.  Used only to pin oracle/lz4f_oracle.c by differential fuzzing
(tests/test_oracle_liblz4.py) and, optionally, as an extra reported CPU figure in
bench.py.  liblz4 is the third-party dependency behind ``lz4framed.compress``
(ref:snacc/pairwise_ncd.py:12,80); it is not part of /root/reference.
"""
import ctypes
import os

import numpy as np

_CANDIDATES = (
    "/opt/conda/lib/liblz4.so.1.9.3",
    "/usr/lib/x86_64-linux-gnu/liblz4.so.1.9.3",
    "/opt/conda/lib/liblz4.so.1",
    "/usr/lib/x86_64-linux-gnu/liblz4.so.1",
)
_lib = None


def available():
    return _load() is not None


def _load():
    global _lib
    if _lib is None:
        for p in _CANDIDATES:
            if os.path.exists(p):
                try:
                    L = ctypes.CDLL(p)
                    L.LZ4_versionString.restype = ctypes.c_char_p
                    L.LZ4F_compressFrameBound.restype = ctypes.c_size_t
                    L.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
                    L.LZ4F_compressFrame.restype = ctypes.c_size_t
                    L.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                     ctypes.c_size_t, ctypes.c_void_p]
                    L.LZ4F_isError.restype = ctypes.c_uint
                    L.LZ4F_isError.argtypes = [ctypes.c_size_t]
                    _lib = L
                    break
                except OSError:
                    continue
    return _lib


def version():
    L = _load()
    return L.LZ4_versionString().decode() if L else None


def compress_frame(data):
    """LZ4F_compressFrame(dst, cap, src, n, NULL) -> bytes (what lz4framed.compress returns)."""
    L = _load()
    if L is None:
        raise RuntimeError("liblz4 not available")
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) \
        else np.ascontiguousarray(data, dtype=np.uint8)
    n = int(a.size)
    cap = L.LZ4F_compressFrameBound(n, None)
    dst = ctypes.create_string_buffer(cap)
    r = L.LZ4F_compressFrame(dst, cap, a.ctypes.data if n else None, n, None)
    if L.LZ4F_isError(r):
        raise RuntimeError("LZ4F_compressFrame error")
    return dst.raw[:r]


def frame_size(data):
    return len(compress_frame(data))
