"""Dev aid: end-to-end `snacc <dir> -c CODEC` on N synthetic FASTA files (wall time by phase): gpu_cli_e2e.py N L [lz4|gzip|zlib]"""
import sys, time, os, tempfile, shutil
sys.path.insert(0, '.')
from pathlib import Path
import numpy as np, torch
from bench import lcg_genomes_torch
N, L = int(sys.argv[1]), int(sys.argv[2])
CODEC = sys.argv[3] if len(sys.argv) > 3 else "lz4"
d = Path(tempfile.mkdtemp(dir="/tmp")); fa = d / "fa"; fa.mkdir()
t = time.time()
for i, g in enumerate(lcg_genomes_torch(N, L, 1, torch.device("cuda", 0))):
    rows = np.frombuffer(g.tobytes()[: L // 80 * 80], dtype=np.uint8).reshape(-1, 80)
    body = np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).tobytes() + g.tobytes()[L // 80 * 80:] + b"\n"
    (fa / f"g{i:04d}.fasta").write_bytes(b">g%d\n" % i + body)
print(f"wrote {N} FASTA files in {time.time()-t:.1f}s", flush=True)
from click.testing import CliRunner
from snacc_amd import cli as C
os.chdir(d)
marks = {}
orig_up = C.gpu_matrix
def timed(*a, **k):
    t0 = time.time(); m = orig_up(*a, **k); marks["lz4_matrix"] = time.time() - t0; return m
C.gpu_matrix = timed
t0 = time.time()
res = CliRunner().invoke(C.cli, [str(fa), "-o", "out.csv", "-c", CODEC, "--no-show-progress"])
tot = time.time() - t0
print("exit", res.exit_code, res.output[-200:] if res.exit_code else "")
print(f"CLI total {tot:.2f}s; ingest+upload+singles+pairs+NCD {marks.get('lz4_matrix', 0):.2f}s; CSV+log {tot - marks.get('lz4_matrix', 0):.2f}s; csv bytes {os.path.getsize('out.csv')}")
shutil.rmtree(d)
