"""Dev aid: where does host time go in pairs_device?"""
import sys, time
sys.path.insert(0, '.')
import torch
from bench import lcg_genomes_torch
from snacc_amd.hip_backend import HipContext
N, L, R = 1024, 1000000, 84
dev = torch.device('cuda', 0)
seqs = lcg_genomes_torch(N, L, 1, dev)
ctx = HipContext(0); ctx.upload(seqs)
tile = torch.zeros((R, N), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream()
for k in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.pairs_device(k * R, (k + 1) * R, tile.data_ptr(), st.cuda_stream)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"call {1e3*(t1-t0):.1f} ms, wait {1e3*(t2-t1):.1f} ms", flush=True)
ctx.sync(st.cuda_stream)
print("lib ms", ctx.last_pairs_ms())
