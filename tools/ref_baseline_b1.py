#!/usr/bin/env python3
"""the reference as it is -- `snacc <dir> -c lz4 -n T` (ThreadPoolExecutor over
compressed_size, ref:snacc/cli.py:104-136) -- timed in THIS container on a sub-sample.

The reference's Python and its two absent third-party modules cannot go to the GPU box: the
unmodified reference is imported from /root/reference with the same in-memory stand-ins
tests/golden/make_golden.py uses (lz4framed.compress -> liblz4 1.9.3 LZ4F_compressFrame(NULL prefs),
a minimal Bio.SeqIO.parse).  Reports ordered pairs/s and NCD/s for T in {1, nproc}.

    python tools/ref_baseline_b1.py [N=16] [L=1000000]       # N files of L bases (LCG genomes, seed 1 + index)
"""
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
import make_golden  # noqa: E402  (stand-ins + generator; imports nothing of the reference by itself)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000


def main():
    import oracle
    make_golden.install_standins()
    sys.path.insert(0, str(make_golden.REF))
    import snacc.cli as rcli                       # the reference, unmodified
    from click.testing import CliRunner
    cores = len(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory() as td:
        d = Path(td) / "fa"
        d.mkdir()
        for g in range(N):
            make_golden.write_fasta(d / f"g{g:04d}.fasta", [(f"g{g}", bytes(oracle.lcg_genome(1 + g, L)).decode())])
        for threads in sorted({1, cores}):
            out = Path(td) / f"out_{threads}.csv"
            cwd = os.getcwd()
            os.chdir(td)
            t0 = time.perf_counter()
            try:
                res = CliRunner().invoke(rcli.cli, [str(d), "-o", str(out), "-c", "lz4", "-n", str(threads),
                                                    "--no-show-progress", "--no-log"])
            finally:
                os.chdir(cwd)
            dt = time.perf_counter() - t0
            assert res.exit_code == 0, res.output
            pairs = N * N + N
            print(f"B1 reference CLI, {N} x {L} bp, -c lz4 -n {threads} ({cores} usable cores): {dt:.2f} s wall, "
                  f"{pairs / dt:.1f} compressions/s, {N * (N + 1) / 2 / dt:.1f} NCD/s, {dt / (N * N) * 1e3:.1f} ms per ordered pair",
                  flush=True)


if __name__ == "__main__":
    main()
