"""NCD matrix assembly on the host (float64), ref:snacc/cli.py:131-136 + pairwise_ncd.py:93-111.

Floats are never computed on the GPU: the kernels return integer frame sizes and the
formula below reproduces the reference's Python arithmetic bit for bit (int -> float64
true division is correctly rounded in both, and division by a positive number is
monotonic, so min-then-divide equals divide-then-min)."""
import sys

import numpy as np

#: ``sys.getsizeof(b"")`` -- the constant the reference adds to every size by measuring the
#: bytes *object* (ref:snacc/pairwise_ncd.py:90).  33 on 64-bit CPython.
GETSIZEOF_OVERHEAD = sys.getsizeof(b"")


def ncd_matrix(singles, pairs):
    """singles: (N,) compressed sizes C(x_i); pairs: (N, N) with pairs[i, j] = C(x_i + x_j).
    Sizes must already include the getsizeof overhead.  Returns the (N, N) float64 matrix
    D[i, j] = compute_distance(C_i, C_j, C_ij, C_ji)."""
    s = np.asarray(singles, dtype=np.int64)
    p = np.asarray(pairs, dtype=np.int64)
    # (few passes over the N x N array, in place: the assembly is part of the measured matrix wall time)
    num = np.minimum(p, p.T)
    num -= np.minimum(s[:, None], s[None, :])
    out = num.astype(np.float64)
    out /= np.maximum(s[:, None], s[None, :]).astype(np.float64)
    return out


def ncd_matrix_raw(singles, pairs, overhead=GETSIZEOF_OVERHEAD, threads=None):
    """The same matrix from the RAW uint32 sizes the kernels return (``overhead`` = what every size grows by on the
    way to the reference's numbers: ``sys.getsizeof(b"")``), computed by the library's host threads
    (``snk_ncd_matrix_u32``: 0.09 s of numpy passes over a 1024 x 1024 matrix become a few ms -- the assembly is part of the
    measured matrix wall time and of every rank's serial share).  Without the library, or for other dtypes, the numpy
    statement above; the CPU tests hold the two bit-equal."""
    s = np.asarray(singles)
    p = np.asarray(pairs)
    n = len(s)
    if s.dtype == np.uint32 and p.dtype == np.uint32 and p.shape == (n, n) and n > 0:
        try:
            from . import hip_backend
            lib = hip_backend.load()
        except Exception:                                        # noqa: BLE001  (no library: the numpy statement)
            lib = None
        if lib is not None:
            s = np.ascontiguousarray(s)
            p = np.ascontiguousarray(p)
            out = np.empty((n, n), dtype=np.float64)
            rc = lib.snk_ncd_matrix_u32(s.ctypes.data, p.ctypes.data, n, int(overhead), out.ctypes.data,
                                        threads or hip_backend.default_threads())
            if rc == 0:
                return out
    return ncd_matrix(s.astype(np.int64) + int(overhead), p.astype(np.int64) + int(overhead))
