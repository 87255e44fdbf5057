"""ctypes binding to ``libsnacc_hip.so`` (C-ABI in ``include/snacc_hip.h``).

This is the product path for ``-c lz4``: it replaces the per-item
``lz4framed.compress`` calls of ref:snacc/pairwise_ncd.py:80 with batched HIP
kernels on an MI355X.  There is deliberately no CPU fallback: if the shared
library is missing or no GPU is visible, every entry point raises
:class:`HipBackendError`.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
#: the in-tree library; SNACC_HIP_LIB points the binding at another build of it (development A/B runs)
LIB_PATH = os.environ.get("SNACC_HIP_LIB") or os.path.join(_HERE, "libsnacc_hip.so")
ABI_VERSION = 1

#: every symbol ``include/snacc_hip.h`` declares (checked by the CPU test-suite)
EXPORTS = (
    "snk_version", "snk_last_error", "snk_ctx_create", "snk_ctx_destroy", "snk_set_option",
    "snk_upload", "snk_num_sequences", "snk_lengths", "snk_num_packed", "snk_fast_chains", "snk_num_compact_hashes", "snk_singles", "snk_pairs",
    "snk_singles_rows", "snk_upload_times", "snk_ncd_matrix_u32",
    "snk_pairs_device", "snk_pairs_list", "snk_frames_list", "snk_sync", "snk_last_pairs_ms", "snk_pairs_ms_log",
    "snk_fasta_extract", "snk_fasta_extract_many", "snk_fasta_last_error", "snk_free", "snk_upload_fasta", "snk_csv_rows_f64",
    "snk_deflate_prepare", "snk_deflate_singles", "snk_deflate_pairs", "snk_deflate_pairs_list", "snk_deflate_pairs_device", "snk_deflate_last_ms",
)

#: deflate level and wrapper bytes behind the reference's gzip / zlib choices
#: (ref:snacc/pairwise_ncd.py:73-78: gzip.compress -> level 9 + 18 B, zlib.compress -> level 6 + 6 B)
DEFLATE = {"gzip": (9, 18), "zlib": (6, 6)}


class HipBackendError(RuntimeError):
    pass


class ArenaTooBig(HipBackendError):
    """The sequences of one upload exceed the 32-bit offset range of the device arenas (about 4.29 GB of
    residues): the caller processes the matrix in blocks of sequences (snacc_amd.cli.blocked_sizes)."""


_lib = None


def build(force=False):
    """Compile the library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h", ".cpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "snacc_hip.h"))
    stale = not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", src_dir])
    return LIB_PATH


def load():
    """dlopen the library and declare signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipBackendError(
            f"{LIB_PATH} not found: build it with `make -C snacc_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "The lz4 path has no CPU fallback.")
    try:
        # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  A process must end
        # up with ONE HIP runtime: if torch is installed, load it first so that this library's
        # NEEDED libamdhip64.so.7 resolves to the copy torch uses (the other order leaves two
        # runtimes and hipGetDeviceCount() then reports no device).
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipBackendError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, u32p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    L.snk_version.restype = i32
    L.snk_version.argtypes = []
    L.snk_last_error.restype = ctypes.c_char_p
    L.snk_last_error.argtypes = [vp]
    L.snk_ctx_create.restype = i32
    L.snk_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.snk_ctx_destroy.restype = None
    L.snk_ctx_destroy.argtypes = [vp]
    L.snk_set_option.restype = i32
    L.snk_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_long]
    L.snk_upload.restype = i32
    L.snk_upload.argtypes = [vp, i32, vp, vp]
    L.snk_num_sequences.restype = i32
    L.snk_num_sequences.argtypes = [vp]
    L.snk_lengths.restype = i32
    L.snk_lengths.argtypes = [vp, vp]
    L.snk_num_packed.restype = i32
    L.snk_num_packed.argtypes = [vp]
    L.snk_fast_chains.restype = i32
    L.snk_fast_chains.argtypes = [vp]
    L.snk_num_compact_hashes.restype = i32
    L.snk_num_compact_hashes.argtypes = [vp]
    L.snk_singles.restype = i32
    L.snk_singles.argtypes = [vp, u32p]
    if hasattr(L, "snk_singles_rows"):          # (absent from the round-3 builds that development A/B runs load through SNACC_HIP_LIB)
        L.snk_singles_rows.restype = i32
        L.snk_singles_rows.argtypes = [vp, i32, i32, u32p]
        L.snk_upload_times.restype = i32
        L.snk_upload_times.argtypes = [vp, vp, i32]
        L.snk_ncd_matrix_u32.restype = i32
        L.snk_ncd_matrix_u32.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_uint32, vp, i32]
    L.snk_pairs.restype = i32
    L.snk_pairs.argtypes = [vp, i32, i32, u32p]
    L.snk_pairs_device.restype = i32
    L.snk_pairs_device.argtypes = [vp, i32, i32, vp, vp]
    L.snk_pairs_list.restype = i32
    L.snk_pairs_list.argtypes = [vp, i32, vp, u32p]
    L.snk_frames_list.restype = i32
    L.snk_frames_list.argtypes = [vp, i32, vp, vp, vp]
    L.snk_sync.restype = i32
    L.snk_sync.argtypes = [vp, vp]
    L.snk_last_pairs_ms.restype = ctypes.c_double
    L.snk_last_pairs_ms.argtypes = [vp]
    L.snk_pairs_ms_log.restype = i32
    L.snk_pairs_ms_log.argtypes = [vp, ctypes.POINTER(ctypes.c_double), i32]
    L.snk_fasta_extract.restype = i32
    L.snk_fasta_extract.argtypes = [ctypes.c_char_p, i32, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint64)]
    L.snk_fasta_extract_many.restype = i32
    L.snk_fasta_extract_many.argtypes = [i32, vp, i32, i32, vp, vp]
    L.snk_fasta_last_error.restype = ctypes.c_char_p
    L.snk_fasta_last_error.argtypes = []
    L.snk_free.restype = None
    L.snk_free.argtypes = [vp]
    L.snk_upload_fasta.restype = i32
    L.snk_upload_fasta.argtypes = [vp, i32, vp, i32, i32]
    L.snk_csv_rows_f64.restype = i32
    L.snk_csv_rows_f64.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, vp, ctypes.c_uint64, vp, i32]
    L.snk_deflate_prepare.restype = i32
    L.snk_deflate_prepare.argtypes = [vp, i32]
    L.snk_deflate_singles.restype = i32
    L.snk_deflate_singles.argtypes = [vp, i32, vp]
    L.snk_deflate_pairs.restype = i32
    L.snk_deflate_pairs.argtypes = [vp, i32, i32, i32, vp]
    L.snk_deflate_pairs_list.restype = i32
    L.snk_deflate_pairs_list.argtypes = [vp, i32, i32, vp, vp]
    L.snk_deflate_pairs_device.restype = i32
    L.snk_deflate_pairs_device.argtypes = [vp, i32, i32, i32, vp, vp]
    L.snk_deflate_last_ms.restype = ctypes.c_double
    L.snk_deflate_last_ms.argtypes = [vp]
    if L.snk_version() != ABI_VERSION:
        raise HipBackendError(f"ABI mismatch: library {L.snk_version()}, binding {ABI_VERSION}")
    _lib = L
    return L


def _as_u8(seq):
    if isinstance(seq, np.ndarray):
        return np.ascontiguousarray(seq, dtype=np.uint8)
    if isinstance(seq, str):
        seq = seq.encode("utf-8")  # ref:snacc/pairwise_ncd.py:69  bytes(sequence, encoding="utf-8")
    return np.frombuffer(seq, dtype=np.uint8)


E_TOOBIG, E_EMPTY, E_MIXED = -4, -6, -7


def _raise_fasta(L, rc):
    msg = L.snk_fasta_last_error().decode(errors="replace")
    if rc in (E_EMPTY, E_MIXED):
        raise ValueError(msg)              # the reference's / Biopython's exception type
    if rc == E_TOOBIG and "arena" in msg:
        raise ArenaTooBig(f"upload failed [{rc}]: {msg}")
    raise HipBackendError(f"FASTA ingest failed [{rc}]: {msg}")


def fasta_extract(path, reverse_complement=False):
    """Native ``extract_sequences`` for one file -> ``bytes`` (host only, no GPU needed)."""
    L = load()
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    rc = L.snk_fasta_extract(os.fsencode(str(path)), int(bool(reverse_complement)), ctypes.byref(out), ctypes.byref(n))
    if rc != 0:
        _raise_fasta(L, rc)
    try:
        return ctypes.string_at(out, n.value)
    finally:
        L.snk_free(out)


def default_threads():
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 32))
    except AttributeError:
        return max(1, min(os.cpu_count() or 1, 32))


class HipContext:
    """One device context (one per process / GPU).  Not thread-safe."""

    def __init__(self, device=0, **options):
        self._L = load()
        h = ctypes.c_void_p()
        rc = self._L.snk_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise HipBackendError(f"snk_ctx_create({device}) failed [{rc}]: "
                                  f"{self._L.snk_last_error(None).decode()}")
        self._h = h
        self.device = int(device)
        self.n = 0
        # SNACC_LZ4_CONTENT_SIZE=1: LZ4 frames carry the 8-byte content-size field (every lz4 size
        # grows by 8, emitted frames get the field).  liblz4's NULL-preferences frame (the default)
        # has none; whether a given py-lz4framed wheel sets it is not knowable offline.
        if os.environ.get("SNACC_ARENA_LIMIT") and "arena_limit" not in options:      # tests: force the blocked path
            options = dict(options, arena_limit=int(os.environ["SNACC_ARENA_LIMIT"]))
        if os.environ.get("SNACC_LZ4_CONTENT_SIZE", "0") not in ("", "0"):
            options = dict(content_size=1, **{k: v for k, v in options.items() if k != "content_size"}) \
                if "content_size" not in options else options
        for k, v in options.items():
            self.set_option(k, v)

    # -- plumbing -----------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            err = ArenaTooBig if (rc == E_TOOBIG and "arena" in self._L.snk_last_error(self._h).decode()) else HipBackendError
            raise err(f"{what} failed [{rc}]: {self._L.snk_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.snk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key, value):
        self._check(self._L.snk_set_option(self._h, key.encode(), int(value)), f"snk_set_option({key})")

    # -- data ---------------------------------------------------------------------------
    def upload(self, sequences):
        """sequences: iterable of bytes / str / uint8 arrays (the extracted residues per file)."""
        arrs = [_as_u8(s) for s in sequences]
        n = len(arrs)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (ctypes.c_uint64 * max(n, 1))(*[a.size for a in arrs])
        self._check(self._L.snk_upload(self._h, n, ptrs, lens), "snk_upload")
        self.n = n
        return self

    def upload_fasta(self, paths, reverse_complement=False, threads=None):
        """Parse every FASTA file once on host threads (C++) and upload; the sequence bytes never
        become Python objects.  Raises ValueError like the reference for empty/malformed files."""
        paths = [os.fsencode(str(p)) for p in paths]
        n = len(paths)
        arr = (ctypes.c_char_p * max(n, 1))(*paths)
        rc = self._L.snk_upload_fasta(self._h, n, arr, int(bool(reverse_complement)), threads or default_threads())
        if rc != 0:
            _raise_fasta(self._L, rc)
        self.n = n
        return self

    def lengths(self):
        """Lengths of the resident sequences (uint64 array)."""
        out = np.zeros(self.n, dtype=np.uint64)
        self._check(self._L.snk_lengths(self._h, out.ctypes.data), "snk_lengths")
        return out

    @property
    def num_packed(self):
        return self._L.snk_num_packed(self._h)

    def fast_chains(self):
        """Ordered pairs one workgroup of the 2-bit kernel holds in flight (lanes x waves) under the current options."""
        rc = self._L.snk_fast_chains(self._h)
        if rc < 0:
            self._check(rc, "snk_fast_chains")
        return rc

    @property
    def num_compact_hashes(self):
        """> 0 when the byte kernel runs with its compact table (that many distinct 5-byte hashes)."""
        return self._L.snk_num_compact_hashes(self._h)

    # -- phase A / B --------------------------------------------------------------------
    def singles(self):
        out = np.zeros(self.n, dtype=np.uint32)
        self._check(self._L.snk_singles(self._h, out.ctypes.data), "snk_singles")
        return out

    def singles_rows(self, row_begin, row_end):
        """Single sizes of the sequences [row_begin, row_end) (with ``defer_singles``: their phase A runs now)."""
        out = np.zeros(max(row_end - row_begin, 0), dtype=np.uint32)
        self._check(self._L.snk_singles_rows(self._h, row_begin, row_end, out.ctypes.data), "snk_singles_rows")
        return out

    def upload_times(self):
        """Host wall time (s) of the last upload's stages (``snk_upload_times``)."""
        buf = (ctypes.c_double * 7)()
        self._L.snk_upload_times(self._h, buf, 7)
        keys = ("h2d", "classify", "pack", "hashsets_slotstream", "tables", "singles", "total")
        return {k: buf[i] * 1e-3 for i, k in enumerate(keys)}

    def pairs(self, row_begin=0, row_end=None):
        row_end = self.n if row_end is None else row_end
        out = np.zeros((max(row_end - row_begin, 0), self.n), dtype=np.uint32)
        self._check(self._L.snk_pairs(self._h, row_begin, row_end, out.ctypes.data), "snk_pairs")
        return out

    def pairs_device(self, row_begin, row_end, d_ptr, stream=None):
        """Async launch; writes (row_end-row_begin)*n uint32 at device address ``d_ptr``."""
        self._check(self._L.snk_pairs_device(self._h, row_begin, row_end, ctypes.c_void_p(d_ptr),
                                             ctypes.c_void_p(stream) if stream else None),
                    "snk_pairs_device")

    def pairs_list(self, ij):
        ij = np.ascontiguousarray(ij, dtype=np.int32).reshape(-1, 2)
        out = np.zeros(len(ij), dtype=np.uint32)
        self._check(self._L.snk_pairs_list(self._h, len(ij), ij.ctypes.data, out.ctypes.data), "snk_pairs_list")
        return out

    def frames(self, items):
        """Compressed LZ4 frames (bytes) of ``items``: ``(i, -1)`` = sequence i alone, ``(i, j)`` = seq_i + seq_j."""
        ij = np.ascontiguousarray(items, dtype=np.int32).reshape(-1, 2)
        n = len(ij)
        if n == 0:
            return []
        sizes = np.zeros(n, dtype=np.uint64)
        single = ij[:, 1] < 0
        if single.any():
            sizes[single] = self.singles()[ij[single, 0]]
        if (~single).any():
            sizes[~single] = self.pairs_list(ij[~single])
        offsets = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(sizes, out=offsets[1:])
        out = np.zeros(int(offsets[-1]), dtype=np.uint8)
        self._check(self._L.snk_frames_list(self._h, n, ij.ctypes.data, offsets.ctypes.data, out.ctypes.data),
                    "snk_frames_list")
        return [out[int(offsets[t]):int(offsets[t + 1])].tobytes() for t in range(n)]

    # ---- gzip / zlib sizes (len(gzip.compress(..)) / len(zlib.compress(..)), wrapper included) ----
    def deflate_singles(self, algorithm):
        level, wrapper = DEFLATE[algorithm]
        out = np.zeros(self.n, dtype=np.uint32)
        self._check(self._L.snk_deflate_singles(self._h, level, out.ctypes.data), "snk_deflate_singles")
        return out + np.uint32(wrapper)

    def deflate_pairs(self, algorithm, row_begin=0, row_end=None):
        level, wrapper = DEFLATE[algorithm]
        row_end = self.n if row_end is None else row_end
        out = np.zeros((row_end - row_begin, self.n), dtype=np.uint32)
        self._check(self._L.snk_deflate_pairs(self._h, level, row_begin, row_end, out.ctypes.data), "snk_deflate_pairs")
        return out + np.uint32(wrapper)

    def deflate_pairs_list(self, algorithm, ij):
        level, wrapper = DEFLATE[algorithm]
        ij = np.ascontiguousarray(ij, dtype=np.int32).reshape(-1, 2)
        out = np.zeros(len(ij), dtype=np.uint32)
        self._check(self._L.snk_deflate_pairs_list(self._h, level, len(ij), ij.ctypes.data, out.ctypes.data),
                    "snk_deflate_pairs_list")
        return out + np.uint32(wrapper)

    def deflate_pairs_device(self, algorithm, row_begin, row_end, d_ptr, stream=None):
        """Asynchronous rows [row_begin, row_end): RAW deflate stream sizes (no wrapper bytes) as u32 to device
        memory at `d_ptr`, launched on `stream` (a hipStream_t as int; None = the context's stream)."""
        level, _ = DEFLATE[algorithm]
        self._check(self._L.snk_deflate_pairs_device(self._h, level, row_begin, row_end, ctypes.c_void_p(d_ptr),
                                                     ctypes.c_void_p(stream) if stream else None), "snk_deflate_pairs_device")

    def deflate_last_ms(self):
        return float(self._L.snk_deflate_last_ms(self._h))

    def sync(self, stream=None):
        self._check(self._L.snk_sync(self._h, ctypes.c_void_p(stream) if stream else None), "snk_sync")

    def last_pairs_ms(self):
        return float(self._L.snk_last_pairs_ms(self._h))

    def pairs_ms_log(self, cap=4096):
        """Device times (ms) of the pair launches since the previous call (after sync); clears the log."""
        buf = (ctypes.c_double * cap)()
        k = self._L.snk_pairs_ms_log(self._h, buf, cap)
        if k < 0:
            self._check(k, "snk_pairs_ms_log")
        return [float(buf[i]) for i in range(min(k, cap))]
