"""``snacc`` command line (drop-in for ref:snacc/cli.py) with ``-c lz4`` on MI355X.

Flags, prompts, banners, CSV layout and the Markdown run log are the reference's
(ref:snacc/cli.py:16-68, 138-160, 180-194).  What differs is *how* the sizes are obtained:

* ``-c lz4``: every input file is parsed ONCE, all sequences are uploaded to the GPU and the
  N + N*N frame sizes come from two batched calls into ``libsnacc_hip.so`` (the reference
  re-parses both files and calls ``lz4framed.compress`` for each of the N*N tasks,
  ref:snacc/cli.py:120-129).  Under ``torchrun`` (WORLD_SIZE > 1) the rows are sharded over
  the ranks and assembled with one RCCL all-gather; rank 0 writes the outputs.
* other codecs: the reference's own thread-pool flow over ``compressed_size`` (CPU, stdlib).
"""
import concurrent.futures
import itertools
import os
import sys
from datetime import datetime
from pathlib import Path

import click
import numpy as np
import pandas as pd
from tqdm import tqdm

from .matrix import GETSIZEOF_OVERHEAD, ncd_matrix, ncd_matrix_raw
from .pairwise_ncd import blob_name, compressed_size, compute_distance, extract_sequences
from .version import __version__

FASTA_SUFFIXES = [".fasta", ".fna", ".fa", ".faa", ".fsa"]

#: stands in for ``lz4framed.__version__`` in the run log (ref:snacc/cli.py:152)
LZ4_BACKEND_VERSION = f"n/a (snacc_amd {__version__} HIP backend; liblz4 1.9.3 LZ4F_compressFrame semantics)"


def discover_files(sequences, fasta=(), directories=()):
    """Input discovery of ref:snacc/cli.py:90-102: explicit files + FASTA-suffixed entries of the
    given directories (non-recursive), de-duplicated, sorted by absolute path string."""
    sequences = [Path(sequence) for sequence in sequences]
    files = [path for path in sequences if path.is_file()]
    files.extend([Path(_f) for _f in fasta])
    sequences.extend([Path(_f) for _f in directories])
    for directory in [path for path in sequences if path.is_dir()]:
        for f in directory.iterdir():
            if f.suffix.lower() in FASTA_SUFFIXES:
                files.append(f)
    return sorted(list(set(files)), key=lambda x: str(x.absolute()))


def write_matrix_csv(files, matrix, output):
    """CSV exactly as ``DataFrame.pivot(index='file', columns='file2', values='ncd').to_csv``
    writes it (ref:snacc/cli.py:138-142): header ``file,<path>...``, one row per file, rows and
    columns in the sort order pandas gives ``Path`` objects, floats in shortest round-trip form,
    minimal quoting, ``os.linesep`` line ends.  Written directly: building the
    reference's 1 M-row long-form DataFrame and pivoting it is the slow part at N >= 1024;
    ``tests/test_host_logic.py`` checks byte equality with the pandas route."""
    order = sorted(range(len(files)), key=lambda i: files[i])
    labels = [str(files[i]) for i in order]
    m = np.ascontiguousarray(np.asarray(matrix, dtype=np.float64)[np.ix_(order, order)])
    fmt = _csv_row_formatter()
    if fmt is None:
        return write_matrix_csv_python(files, matrix, output)
    import csv
    import io
    sio = io.StringIO()
    wr = csv.writer(sio, lineterminator=os.linesep, quoting=csv.QUOTE_MINIMAL)
    wr.writerow(["file"] + labels)
    eol = os.linesep.encode()
    try:
        with open(output, "wb") as f:
            f.write(sio.getvalue().encode("utf-8"))
            # the float fields come from the library CSV_CHUNK_ROWS rows at a time, into one reused buffer: the text of the
            # whole matrix (25 bytes per value) is never held at once
            for r0 in range(0, len(labels), CSV_CHUNK_ROWS):
                bodies = fmt(m[r0:r0 + CSV_CHUNK_ROWS])
                if bodies is None:
                    raise MemoryError("snk_csv_rows_f64 failed")
                for label, body in zip(labels[r0:r0 + CSV_CHUNK_ROWS], bodies):
                    sio.seek(0); sio.truncate()
                    wr.writerow([label, ""])                 # "label," with the quoting the csv module gives the label
                    f.write(sio.getvalue()[: -len(os.linesep)].encode("utf-8"))
                    f.write(body)
                    f.write(eol)
    except MemoryError:                                      # no room for the staging buffer: the row-by-row Python statement
        return write_matrix_csv_python(files, matrix, output)


#: rows of the matrix formatted per call of the library (one reused staging buffer of rows x cols x 25 bytes)
CSV_CHUNK_ROWS = 256


def _csv_row_formatter():
    """A function `fmt(rows_of_m) -> list of bytes`: the float fields of each row joined by ',', formatted by the library's
    host code (Python's repr of a float, NaN as an empty field: `snk_csv_rows_f64`) -- formatting a million floats in
    Python costs more than a sixth of the whole `snacc <1024 genomes> -c lz4` run.  None when the library cannot be loaded
    (the Python statement :func:`write_matrix_csv_python` is then used; the tests hold the two and the pandas route
    equal).  The staging buffer is allocated once, for CSV_CHUNK_ROWS rows, and reused."""
    try:
        from . import hip_backend
        lib = hip_backend.load()
    except Exception:                                        # noqa: BLE001  (no library: the Python statement)
        return None
    state = {}

    def fmt(m):
        rows, cols = m.shape
        if rows == 0 or cols == 0:
            return [b""] * rows
        stride = cols * 25                                   # SNK_CSV_FIELD_MAX per value
        if state.get("cap", 0) < rows * stride:
            state["out"] = np.empty(rows * stride, dtype=np.uint8)
            state["cap"] = rows * stride
        out = state["out"]
        m = np.ascontiguousarray(m)
        lens = np.zeros(rows, dtype=np.uint32)
        rc = lib.snk_csv_rows_f64(m.ctypes.data, rows, cols, out.ctypes.data, stride, lens.ctypes.data,
                                  hip_backend.default_threads())
        if rc != 0:
            return None
        return [out[r * stride: r * stride + int(lens[r])].tobytes() for r in range(rows)]
    return fmt


def _csv_row_bodies(m):
    """The float fields of every row of `m` at once (tests; the writer goes chunk by chunk)."""
    fmt = _csv_row_formatter()
    return None if fmt is None else fmt(np.ascontiguousarray(m))


def write_matrix_csv_python(files, matrix, output):
    """The Python statement of :func:`write_matrix_csv` (csv module + ``repr``); its checker, and the route without the library."""
    import csv
    order = sorted(range(len(files)), key=lambda i: files[i])
    labels = [str(files[i]) for i in order]
    m = np.asarray(matrix, dtype=np.float64)[np.ix_(order, order)]
    with open(output, "w", newline="", encoding="utf-8") as f:
        wr = csv.writer(f, lineterminator=os.linesep, quoting=csv.QUOTE_MINIMAL)
        wr.writerow(["file"] + labels)
        for label, row in zip(labels, m.tolist()):
            wr.writerow([label] + ["" if v != v else repr(v) for v in row])


def write_matrix_csv_pandas(files, matrix, output):
    """The pandas route (what the reference does after its pivot); kept as the checker for
    :func:`write_matrix_csv`."""
    order = sorted(range(len(files)), key=lambda i: files[i])
    labels = [files[i] for i in order]
    m = np.asarray(matrix)[np.ix_(order, order)]
    df = pd.DataFrame(m, index=pd.Index(labels, name="file", dtype=object),
                      columns=pd.Index(labels, name="file2", dtype=object))
    df.to_csv(output)


def _dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    return world, rank


def save_lz4_blobs(ctx, files, save_directory, rows=None):
    """-s/--save-compression with lz4: write the frame of every file and of every ordered pair,
    named as the reference names them (ref:snacc/pairwise_ncd.py:82-88).  Frames are emitted on the
    GPU in bounded batches."""
    n = len(files)
    items = [(i, -1) for i in range(n)]
    r0, r1 = rows if rows is not None else (0, n)
    items += [(i, j) for i in range(r0, r1) for j in range(n)]
    budget = 256 << 20                                             # bytes of frames per batch
    singles = ctx.singles()
    start = 0
    while start < len(items):
        end, size = start, 0
        while end < len(items) and (end == start or size < budget):
            i, j = items[end]
            size += int(singles[i]) + (int(singles[j]) if j >= 0 else 0)      # upper estimate
            end += 1
        for (i, j), blob in zip(items[start:end], ctx.frames(items[start:end])):
            key = files[i] if j < 0 else (files[i], files[j])
            with open(os.path.join(save_directory.absolute(), blob_name(key, ".lz4")), "wb") as f:
                f.write(blob)
        start = end


def lz4_matrix(files, reverse_complement, show_progress, save_directory=None):
    return gpu_matrix(files, "lz4", reverse_complement, show_progress, save_directory)


def _arena_limit():
    return int(os.environ.get("SNACC_ARENA_LIMIT", str(0xFFF00000)))


def blocked_sizes(ctx, files, algorithm, reverse_complement, show_progress=False, world=1, rank=0, all_reduce=None):
    """Singles and all ordered pair sizes of a file set whose residues do not fit one upload (the device
    arenas are addressed with 32-bit offsets: about 4.29 GB of residues, e.g. 1000 bacterial genomes of
    5 Mbp).  The files are cut into groups of at most 45 % of the limit (by file size, an upper bound of
    the residues); every pair of groups is uploaded once and its ordered pairs (both directions) are
    computed from a pair list.  Same sizes as one upload would give (each pair is a function of its two
    sequences only); G(G+1)/2 uploads for G groups.  An upload that still does not fit (arena padding
    under a small limit) is split in halves; a single pair that does not fit cannot be computed at all
    (an lz4 frame of more than 2.1 GB) and ends the run with a message.

    Multi-rank (`world` > 1): the group pairs are dealt round-robin over the ranks -- no data-path
    collective -- and the two result arrays are summed over the ranks at the end (`all_reduce(array)`:
    every entry is written by exactly one rank, the others hold 0)."""
    from .hip_backend import DEFLATE, ArenaTooBig
    n = len(files)
    sizes = [os.path.getsize(f) + 192 for f in files]            # >= residues + arena padding
    cap = _arena_limit() * 45 // 100
    groups, cur, tot = [], [], 0
    for i in range(n):
        if cur and tot + sizes[i] > cap:
            groups.append(cur)
            cur, tot = [], 0
        cur.append(i)
        tot += sizes[i]
    if cur:
        groups.append(cur)
    deflate = algorithm in DEFLATE
    singles = np.zeros(n, dtype=np.int64)
    pairs = np.zeros((n, n), dtype=np.int64)

    def block(ga, gb):
        """Ordered pairs between the files `ga` and `gb` (both directions), or inside `ga` when gb is None
        (then the singles of `ga` too)."""
        members = ga + (gb or [])
        try:
            ctx.upload_fasta([files[i].absolute() for i in members], reverse_complement=reverse_complement)
        except ArenaTooBig as e:
            big = ga if gb is None or len(ga) >= len(gb) else gb
            if len(big) < 2:
                raise click.ClickException(
                    "the sequences of " + " and ".join(str(files[i]) for i in members) + " do not fit one upload "
                    f"({e}); a pair of this size cannot be compressed on the device") from e
            h1, h2 = big[:len(big) // 2], big[len(big) // 2:]
            if gb is None:
                block(h1, None), block(h2, None), block(h1, h2)
            elif big is ga:
                block(h1, gb), block(h2, gb)
            else:
                block(ga, h1), block(ga, h2)
            return
        na = len(ga)
        if gb is None:
            s = ctx.deflate_singles(algorithm) if deflate else ctx.singles()
            singles[ga] = s.astype(np.int64) + GETSIZEOF_OVERHEAD
            ij = [(i, j) for i in range(na) for j in range(na)]
        else:
            ij = [(i, na + j) for i in range(na) for j in range(len(gb))]
            ij += [(j, i) for i, j in ij]
        got = (ctx.deflate_pairs_list(algorithm, ij) if deflate else ctx.pairs_list(ij)).astype(np.int64)
        for (i, j), v in zip(ij, got):
            pairs[members[i], members[j]] = v + GETSIZEOF_OVERHEAD

    todo = [(a, b) for a in range(len(groups)) for b in range(a, len(groups))][rank::max(world, 1)]
    if show_progress and rank == 0:
        todo = tqdm(todo)
    failure = None
    try:
        for a, b in todo:
            block(groups[a], groups[b] if b != a else None)
    except click.ClickException as e:        # (a pair that fits no upload) -- the other ranks must hear of it before the data reduce
        if world <= 1:
            raise
        failure = e
    if world > 1:
        # agree on failure first: a rank that raised alone would leave the others blocked in the reduce below
        bad = all_reduce(np.array([1 if failure is not None else 0], dtype=np.int64))
        if int(bad[0]):
            raise failure if failure is not None else click.ClickException(
                "another rank could not compute its share of the pairs (see its message); no matrix was written")
        singles, pairs = all_reduce(singles), all_reduce(pairs)
    return singles, pairs


def _sum_over_ranks(backend, device):
    """all_reduce(SUM) of an int64 numpy array over the ranks (device tensors for RCCL, CPU tensors for gloo)."""
    import torch
    import torch.distributed as dist
    from .distributed import _die

    def reduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        if backend == "nccl":
            t = t.to(device)
        try:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        except Exception as e:      # noqa: BLE001
            _die("all_reduce", e)
        return t.cpu().numpy()
    return reduce


def gpu_matrix(files, algorithm, reverse_complement, show_progress, save_directory=None, ctx_factory=None,
               backend="nccl"):
    """Phases A-C for ``-c lz4`` / ``gzip`` / ``zlib`` on the HIP backend.  Returns the float64 NCD
    matrix in `files` order on rank 0 (None on other ranks).

    Under torchrun (WORLD_SIZE > 1) every rank uploads all files, computes its block of rows (blocks of
    equal work, snacc_amd/distributed.py) and takes part in the all-gather; banners, progress bars and
    files are rank 0's.  `ctx_factory` / `backend` exist for the CPU tests (a checker-provided context on
    gloo); the product passes neither.  The multi-rank path has NOT run on RCCL yet."""
    from .hip_backend import DEFLATE

    world, rank = _dist_env()
    chatty = rank == 0
    if chatty:
        click.secho("Compressing individual files...", fg="green")
    n = len(files)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    started_group = False
    if world > 1:
        # rendezvous BEFORE anything touches the GPU; a failure ends the run with status 3 and a message
        import torch.distributed as dist
        from .distributed import init_process_group
        if not dist.is_initialized():
            device = None
            if backend == "nccl":
                import torch
                device = torch.device("cuda", local)
            init_process_group(backend, device)
            started_group = True
    if ctx_factory is None:
        from .hip_backend import HipContext
        ctx_factory = HipContext
    ctx = ctx_factory(local)
    deflate = algorithm in DEFLATE
    if deflate or world > 1:
        # phase A of lz4 (single sizes + prefix snapshots) on demand: a gzip / zlib run never needs it, and a rank of a
        # sharded run computes it for its own rows only (the sizes ride on the tile gathers)
        ctx.set_option("defer_singles", 1)
    try:
        # every file is parsed ONCE, by host threads inside the library, then
        # uploaded; phase A (singles + prefix snapshots) runs as part of the upload
        from .hip_backend import ArenaTooBig
        # File bytes are not residues (headers, line ends, white space): only a set FAR over the limit skips the attempt -- no
        # point in parsing all of it first --, anything else is tried and the library says when it does not fit
        too_big = sum(os.path.getsize(f) for f in files) > _arena_limit() * 3 // 2
        try:
            if not too_big:
                ctx.upload_fasta([f.absolute() for f in files], reverse_complement=reverse_complement)
        except ArenaTooBig:
            too_big = True
        if too_big:
            if save_directory is not None:
                raise click.ClickException("-s/--save-compression on the HIP backend needs the whole set in one upload "
                                           f"(this set does not fit the {_arena_limit() / 1e9:.2f} GB of residues one upload holds)")
            if chatty:
                click.secho("Compressing pairs...", fg="green")
            reduce = None
            if world > 1:
                import torch
                reduce = _sum_over_ranks(backend, torch.device("cuda", local) if backend == "nccl" else None)
            singles, pairs = blocked_sizes(ctx, files, algorithm, reverse_complement, show_progress and chatty,
                                           world=world, rank=rank, all_reduce=reduce)
            return ncd_matrix(singles, pairs) if rank == 0 else None
        singles = None
        if deflate:
            singles = ctx.deflate_singles(algorithm)
        elif world == 1:
            singles = ctx.singles()
        if chatty:
            click.secho("Compressing pairs...", fg="green")
        if world > 1:
            from .distributed import all_pairs_deflate_hip, all_pairs_hip
            if backend == "nccl":
                import torch
                torch.cuda.set_device(ctx.device)
            lengths = ctx.lengths()
            if deflate:
                pairs = all_pairs_deflate_hip(ctx, n, algorithm, lengths=lengths)
            else:
                pairs, singles = all_pairs_hip(ctx, n, lengths=lengths, with_singles=True)
        elif deflate:
            tile = max(1, (1 << 18) // max(n, 1))
            starts = range(0, n, tile)
            if show_progress and n > tile:
                starts = tqdm(starts, total=(n + tile - 1) // tile)
            pairs = (np.concatenate([ctx.deflate_pairs(algorithm, r0, min(n, r0 + tile)) for r0 in starts])
                     if n else np.zeros((0, 0), np.uint32))
        else:
            # row tiles so that --show-progress has something to show on large inputs; results are
            # identical to one call
            tile = max(1, (1 << 21) // max(n, 1))
            starts = range(0, n, tile)
            if show_progress and n > tile:
                starts = tqdm(starts, total=(n + tile - 1) // tile)
            pairs = np.concatenate([ctx.pairs(r0, min(n, r0 + tile)) for r0 in starts]) if n else np.zeros((0, 0), np.uint32)
        if save_directory is not None and rank == 0:
            save_lz4_blobs(ctx, files, save_directory)
    finally:
        ctx.close()
        if started_group:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
    if rank != 0:
        return None
    # raw uint32 sizes -> float64 NCD by the library's host threads (bit-equal to ncd_matrix: tests/test_host_logic.py)
    return ncd_matrix_raw(np.asarray(singles, dtype=np.uint32), np.asarray(pairs, dtype=np.uint32).reshape(n, n))


def threadpool_matrix(files, compression, num_threads, save_compression, reverse_complement, show_progress):
    """The reference's own flow (ref:snacc/cli.py:104-136) for the stdlib codecs."""
    executor = concurrent.futures.ThreadPoolExecutor(max_workers=num_threads)
    click.secho("Compressing individual files...", fg="green")
    compressed_dict = dict(tqdm_parallel_map(
        executor,
        lambda x: compressed_size(sequences=x, algorithm=compression, save_directory=save_compression,
                                  reverse_complement=reverse_complement),
        show_progress, files))
    click.secho("Compressing pairs...", fg="green")
    pair_sizes = dict(tqdm_parallel_map(
        executor,
        lambda x: compressed_size(sequences=x, algorithm=compression, save_directory=save_compression,
                                  reverse_complement=reverse_complement),
        show_progress, itertools.product(compressed_dict.keys(), repeat=2)))
    n = len(files)
    matrix = np.zeros((n, n), dtype=np.float64)
    for i, a in enumerate(files):
        for j, b in enumerate(files):
            matrix[i, j] = compute_distance(compressed_dict[a], compressed_dict[b],
                                            pair_sizes[(a, b)], pair_sizes[(b, a)])
    return matrix


@click.command(context_settings=dict(help_option_names=["-h", "--help"]))
@click.argument("sequences", type=click.Path(exists=True, resolve_path=True), nargs=-1)
@click.option("-f", "--fasta", type=click.Path(dir_okay=False, exists=True, resolve_path=True),
              multiple=True, hidden=True, help="FASTA file containing sequence to compare.")
@click.option("-d", "--directory", "directories",
              type=click.Path(dir_okay=True, file_okay=False, exists=True, resolve_path=True),
              multiple=True, hidden=True, help="Directory containing FASTA files to compare.")
@click.option("-n", "--num-threads", "numThreads", type=int, default=None,
              help="Number of Threads to use (default 5 * number of cores).")
@click.option("-o", "--output", type=click.Path(dir_okay=False, exists=False),
              help="The location for the output CSV file.", required=True,
              # the reference prompts when -o is missing (ref:snacc/cli.py:51-56); under torchrun every rank
              # would prompt, so a multi-rank run requires the option instead
              prompt="Output CSV path" if int(os.environ.get("WORLD_SIZE", "1")) == 1 else False)
@click.option("-s", "--save-compression", "saveCompression",
              type=click.Path(dir_okay=True, file_okay=False, resolve_path=True), default=None,
              help="Save compressed sequence files to the specified directory.")
@click.option("-c", "--compression", default="lzma",
              type=click.Choice(["lzma", "gzip", "bzip2", "zlib", "lz4"]),
              help="The compression algorithm to use. Defaults to lzma.")
@click.option("--show-progress/--no-show-progress", "showProgress", default=True,
              help="Whether to show a progress bar for computing compression distances.")
@click.option("-r", "--reverse_complement", is_flag=True, default=False,
              help="Whether to use the reverse complement of the sequence.")
@click.option("--log/--no-log", "log", default=True, help="Whether to save a log.")
@click.option("--lz4-content-size/--no-lz4-content-size", "lz4ContentSize", default=None,
              help="(not in the reference) lz4 frames carry the 8-byte content-size field, as a py-lz4framed build "
                   "that sets it would produce; default: the SNACC_LZ4_CONTENT_SIZE environment variable, else off.")
def cli(sequences, fasta, directories, numThreads, compression, showProgress, saveCompression, output,
        reverse_complement, log, lz4ContentSize=None):
    start_time = datetime.now()
    if lz4ContentSize is not None:          # read by HipContext when it is created
        os.environ["SNACC_LZ4_CONTENT_SIZE"] = "1" if lz4ContentSize else "0"

    if (fasta or directories) and _dist_env()[1] == 0:
        click.secho("Warning: the -f and -d flags are deprecated and will be removed before release. "
                    "Please pass files and paths directly without the flags.", fg="yellow")
    if saveCompression:
        saveCompression = Path(saveCompression)
    output = Path(output)

    files = discover_files(sequences, fasta, directories)

    # gzip / zlib: on the HIP backend too unless the compressed files themselves are
    # wanted (-s) or the reference's own thread-pool flow is asked for with SNACC_DEFLATE=stdlib
    deflate_on_gpu = (compression in ("gzip", "zlib") and saveCompression is None
                      and os.environ.get("SNACC_DEFLATE", "hip") != "stdlib")
    if compression == "lz4" or deflate_on_gpu:
        matrix = gpu_matrix(files, compression, reverse_complement, showProgress, saveCompression)
    else:
        matrix = threadpool_matrix(files, compression, numThreads, saveCompression, reverse_complement,
                                   showProgress)
    if matrix is None:          # non-zero rank of a multi-GPU run
        return

    write_matrix_csv(files, matrix, output)

    if log:
        rendered = log_template.format(time=datetime.now(),
                                       method=compression,
                                       py_version=str(sys.version.replace("\n", "")),
                                       snacc_version=__version__,
                                       lz4framed_version=LZ4_BACKEND_VERSION,
                                       rev_comp=reverse_complement,
                                       duration=datetime.now() - start_time,
                                       output_path=output.absolute())
        with open(output.stem + ".md", "w") as f:
            print(rendered, file=f)
            for _f in [str(_file.absolute()) for _file in files]:
                print("*", _f, file=f)


def tqdm_parallel_map(executor, fn, showProgress, *iterables, **kwargs):
    """``executor.map`` with a tqdm progress bar, results in completion order
    (ref:snacc/cli.py:162-177)."""
    futures_list = []
    for iterable in iterables:
        futures_list += [executor.submit(fn, i) for i in iterable]
    done = concurrent.futures.as_completed(futures_list)
    if showProgress:
        done = tqdm(done, total=len(futures_list), **kwargs)
    for f in done:
        yield f.result()


log_template = '''# `snacc` Analysis
## Run Information
* Analysis time: {time}
* Analysis duration: {duration}
* Compression method: {method}
* Reverse complement: {rev_comp}
* Output filepath: {output_path}

## Version Information
* Python: {py_version}
* snacc: {snacc_version}
* py-lz4framed: {lz4framed_version}

## Analyzed Files
'''


if __name__ == "__main__":
    cli()
