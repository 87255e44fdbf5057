"""Drop-in for ``snacc.pairwise_ncd`` (ref:snacc/pairwise_ncd.py) with the lz4 codec on MI355X.

Same three functions, same signatures and error behaviour:

* :func:`extract_sequences`  -- ref:snacc/pairwise_ncd.py:15-39
* :func:`compressed_size`    -- ref:snacc/pairwise_ncd.py:42-90
* :func:`compute_distance`   -- ref:snacc/pairwise_ncd.py:93-111

``algorithm == "lz4"`` goes to the HIP backend (``libsnacc_hip.so``) -- there is no CPU
fallback for it.  ``gzip`` and ``zlib`` sizes come from the HIP backend as well (SURVEY.md 8f
N3), unless the compressed bytes themselves are wanted (``save_directory``) or the
environment says ``SNACC_DEFLATE=stdlib`` (the reference's own call, asked for explicitly);
``lzma`` and ``bzip2`` are the stdlib calls the reference makes.  :func:`all_pairs` is the
batched entry the CLI uses instead of N*N single calls (``lz4``, ``gzip``, ``zlib``).
"""
import bz2
import gzip
import lzma
import os
import sys
import threading
import zlib
from pathlib import Path

import numpy as np

from . import fasta
from .matrix import GETSIZEOF_OVERHEAD, ncd_matrix

_EXTENSION = {"lzma": ".lzma", "gzip": ".gz", "bzip2": ".bz2", "zlib": ".ZLIB", "lz4": ".lz4"}

_ctx = None
# One device context serves every single-item call of the process, and a context is not
# thread-safe: upload replaces the resident sequences.  The reference's own usage pattern is a
# ThreadPoolExecutor over compressed_size (ref:snacc/cli.py:104-129), so one call's
# upload + launch + read-back runs under this lock; calls from a pool are serialised, not corrupted.
_ctx_lock = threading.Lock()


def _hip_context():
    """Process-wide device context for single-item calls (device = LOCAL_RANK or 0).
    Callers hold ``_ctx_lock``."""
    global _ctx
    if _ctx is None:
        from .hip_backend import HipContext
        _ctx = HipContext(int(os.environ.get("LOCAL_RANK", "0")))
    return _ctx


def extract_sequences(sequences, reverse_complement=False):
    """Concatenated sequences of a FASTA file, or of two files for a tuple."""
    if type(sequences) == tuple:
        return (extract_sequences(sequences[0], reverse_complement=reverse_complement)
                + extract_sequences(sequences[1], reverse_complement=reverse_complement))
    seq = fasta.read_sequence(sequences.absolute(), reverse_complement)
    if not seq:
        raise ValueError(f"No sequence extracted. Ensure that file {sequences.absolute()} contains a proper "
                         "FASTA definition line (i.e. a line that starts with '>sequence_name').")
    return seq


def compressed_size(sequences, algorithm, reverse_complement=False, save_directory=None, BWT=False, bwte_inputs={}):
    """``(sequences, size)`` where size is what the reference reports: ``sys.getsizeof`` of the
    compressed bytes object (payload length + 33).  ``BWT``/``bwte_inputs`` are accepted and
    ignored, as in the reference."""
    if type(sequences) == tuple:
        parts = [extract_sequences(s, reverse_complement=reverse_complement) for s in sequences]
    else:
        parts = [extract_sequences(sequences, reverse_complement=reverse_complement)]
    file_ext = _EXTENSION[algorithm]          # KeyError for unknown algorithms, as in the reference

    if algorithm == "lz4":
        item = (0, 1) if len(parts) == 2 else (0, -1)
        with _ctx_lock:
            ctx = _hip_context()
            ctx.upload([bytes(p, encoding="utf-8") for p in parts])
            if save_directory:
                # the frame itself, emitted on the GPU (bytes equal liblz4's)
                compressed_seq = ctx.frames([item])[0]
            else:
                n = int(ctx.pairs_list([item])[0]) if len(parts) == 2 else int(ctx.singles()[0])
        if save_directory:
            _save_blob(sequences, save_directory, file_ext, compressed_seq)
            return (sequences, sys.getsizeof(compressed_seq))
        return (sequences, n + GETSIZEOF_OVERHEAD)

    if algorithm in ("gzip", "zlib") and not save_directory and os.environ.get("SNACC_DEFLATE", "hip") != "stdlib":
        item = (0, 1) if len(parts) == 2 else (0, -1)
        with _ctx_lock:
            ctx = _hip_context()
            ctx.upload([bytes(p, encoding="utf-8") for p in parts])
            n = int(ctx.deflate_pairs_list(algorithm, [item])[0])
        return (sequences, n + GETSIZEOF_OVERHEAD)

    sequence = bytes("".join(parts), encoding="utf-8")
    if algorithm == "lzma":
        compressed_seq = lzma.compress(sequence)
    elif algorithm == "gzip":
        compressed_seq = gzip.compress(sequence)
    elif algorithm == "bzip2":
        compressed_seq = bz2.compress(sequence)
    elif algorithm == "zlib":
        compressed_seq = zlib.compress(sequence)
    if save_directory:
        _save_blob(sequences, save_directory, file_ext, compressed_seq)
    return (sequences, sys.getsizeof(compressed_seq))


def blob_name(sequences, file_ext):
    """File name of a saved blob, ref:snacc/pairwise_ncd.py:83-86: ``a.stem + b.name`` for a pair."""
    if type(sequences) == tuple:
        return sequences[0].stem + sequences[1].name + file_ext
    return sequences.name + file_ext


def _save_blob(sequences, save_directory, file_ext, compressed_seq):
    with open(os.path.join(save_directory.absolute(), blob_name(sequences, file_ext)), "wb") as f:
        f.write(compressed_seq)


def compute_distance(x, y, cxy, cyx):
    """Normalized compression distance from four compressed sizes."""
    if x > y:
        return min((cxy - y) / x, (cyx - y) / x)
    elif y > x:
        return min((cxy - x) / y, (cyx - x) / y)
    else:
        return min((cxy - x) / x, (cyx - x) / x)


GPU_ALGORITHMS = ("lz4", "gzip", "zlib")

_STDLIB = {"lzma": lzma.compress, "bzip2": bz2.compress, "gzip": gzip.compress, "zlib": zlib.compress}


def ncd(seq_i, seq_j, compressor="lz4"):
    """Normalized compression distance of two extracted sequences (``str`` or ``bytes``): the sequence-level
    convenience BASELINE.json's north_star names ``compute_distance(seq_i, seq_j, compressor)``.  It is not in the
    reference, whose :func:`compute_distance` takes the four sizes (kept as it is); this wrapper obtains them as
    ref:snacc/pairwise_ncd.py:69-90 does -- ``sys.getsizeof`` of the compressed ``seq``, ``seq_i + seq_j`` and
    ``seq_j + seq_i`` -- lz4 / gzip / zlib on the HIP backend (no CPU fallback), lzma / bzip2 with the stdlib calls
    the reference makes.  ``KeyError`` for an unknown compressor, as ``compressed_size`` raises."""
    _EXTENSION[compressor]
    a, b = (bytes(s, encoding="utf-8") if isinstance(s, str) else bytes(s) for s in (seq_i, seq_j))
    if compressor == "lz4" or (compressor in ("gzip", "zlib") and os.environ.get("SNACC_DEFLATE", "hip") != "stdlib"):
        with _ctx_lock:
            singles, pairs = all_pairs([a, b], compressor, ctx=_hip_context())
        return compute_distance(int(singles[0]), int(singles[1]), int(pairs[0, 1]), int(pairs[1, 0]))
    size = lambda data: sys.getsizeof(_STDLIB[compressor](data))      # noqa: E731
    return compute_distance(size(a), size(b), size(a + b), size(b + a))


def all_pairs(sequences, algorithm="lz4", ctx=None, rows=None):
    """Batched phase A + B of ref:snacc/cli.py:108-129 on the HIP backend.

    sequences: list of ``bytes``/``str`` (already extracted); algorithm: ``lz4``, ``gzip`` or
    ``zlib``.  Returns ``(singles, pairs)`` as int64 arrays *including* the getsizeof overhead
    (and, for gzip / zlib, the 18 / 6 wrapper bytes); ``rows=(r0, r1)`` restricts phase B to a
    row range (multi-GPU sharding)."""
    if algorithm not in GPU_ALGORITHMS:
        raise KeyError(algorithm)
    own = ctx is None
    if own:
        from .hip_backend import HipContext
        ctx = HipContext(int(os.environ.get("LOCAL_RANK", "0")))
    try:
        ctx.upload(sequences)
        r0, r1 = rows if rows is not None else (0, len(sequences))
        if algorithm == "lz4":
            singles, pairs = ctx.singles(), ctx.pairs(r0, r1)
        else:
            singles, pairs = ctx.deflate_singles(algorithm), ctx.deflate_pairs(algorithm, r0, r1)
        singles = singles.astype(np.int64) + GETSIZEOF_OVERHEAD
        pairs = pairs.astype(np.int64) + GETSIZEOF_OVERHEAD
    finally:
        if own:
            ctx.close()
    return singles, pairs


def all_pairs_lz4(sequences, ctx=None, rows=None):
    """:func:`all_pairs` for the lz4 codec."""
    return all_pairs(sequences, "lz4", ctx=ctx, rows=rows)


def ncd_matrix_gpu(sequences, algorithm="lz4", ctx=None):
    """Full N x N NCD matrix (float64) for extracted sequences on one GPU."""
    singles, pairs = all_pairs(sequences, algorithm, ctx=ctx)
    return ncd_matrix(singles, pairs)


def ncd_matrix_lz4(sequences, ctx=None):
    return ncd_matrix_gpu(sequences, "lz4", ctx=ctx)
