import numpy as np, time, sys, os, tempfile
sys.path.insert(0,'.')
from snacc_amd.cli import write_matrix_csv
from pathlib import Path
N=1024
rng=np.random.default_rng(0)
m=rng.random((N,N))*0.1+0.9
files=[Path(f"/tmp/x/genome_{i:04d}.fa") for i in range(N)]
d=tempfile.mkdtemp()
for _ in range(2):
    t=time.perf_counter(); write_matrix_csv(files,m,os.path.join(d,'o.csv')); print('csv',time.perf_counter()-t)
