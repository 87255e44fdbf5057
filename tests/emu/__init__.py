"""Host emulation of one lane of the 2-bit kernel (TEST INFRASTRUCTURE; see snk_host_emu.h)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "..", "snacc_amd", "csrc")
_SO = os.path.join(_HERE, "libfast_emu.so")
_lib = None


def build():
    srcs = [os.path.join(_HERE, f) for f in ("fast_emu.cpp", "snk_host_emu.h")]
    srcs += [os.path.join(_CSRC, f) for f in ("snk_fast.hip.h", "snk_common.hip.h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
                               "-Wno-unused-but-set-variable", "-Wno-maybe-uninitialized", "-DSNK_HOST_EMU", "-I", _HERE, "-I", _CSRC,
                               "-shared", "-fPIC", "-pthread", "-o", _SO, os.path.join(_HERE, "fast_emu.cpp")])
    return _SO


def fast_sizes(seqs, header_bytes=7, exc_limit=128, lower=False, far=False, spec=False):
    """(singles[n], pairs[n, n]) frame sizes the 2-bit kernel's code computes; 0 where the pair is not
    eligible for that kernel (n <= 64 KiB, or a sequence with more non-ACGT places than `exc_limit`
    16-base granules per 2^20 bases (+8) allows; exc_limit=0: pure ACGT only).  lower: the set's letters are acgt
    (the LUTs are then made from liblz4's hashes of the lower-case 5-mers, and upper-case letters are exceptions).
    far: the pairs run as a far chain (table in global memory; sets without exceptions only).
    spec: two lanes per chain (snk_fast_steady_spec, the loop every product launch runs): the emulated lane's partner
    runs on a second host thread, in lockstep."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.emu_fast_sizes.restype = ctypes.c_int
    arrs = [np.ascontiguousarray(np.frombuffer(bytes(s), dtype=np.uint8) if not isinstance(s, np.ndarray) else s,
                                 dtype=np.uint8) for s in seqs]
    n = len(arrs)
    ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data if a.size else 0 for a in arrs])
    lens = (ctypes.c_uint64 * n)(*[int(a.size) for a in arrs])
    singles = np.zeros(n, dtype=np.uint32)
    pairs = np.zeros((n, n), dtype=np.uint32)
    rc = _lib.emu_fast_sizes(n, ptrs, lens, singles.ctypes.data_as(ctypes.c_void_p),
                             pairs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(header_bytes), ctypes.c_uint32(exc_limit), ctypes.c_uint32(1 if lower else 0), ctypes.c_uint32(1 if far else 0),
                             ctypes.c_uint32(1 if spec else 0))
    if rc != 0:
        raise RuntimeError(f"emulated kernel reported status {rc}")
    return singles, pairs



def other_mode_trips():
    """Trips the steady loop's other-case mode has made so far in this process (soft-masked stretches)."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.emu_fast_sizes.restype = ctypes.c_int
    _lib.emu_other_mode_trips.restype = ctypes.c_ulonglong
    return int(_lib.emu_other_mode_trips())
