#!/bin/bash
# Dev aid (GPU box): two lanes per chain (the default, fast_spec=1) against one lane per chain (--opt fast_spec=0) on the three
# data sets of bench.py and on two other launch shapes (256 x 100 kbp; 336 rows per launch): NCD/s and ms per step of each arm.
# Usage: tools/gpu_spec_ab.sh        (run from the repo root on the GPU box)
for d in lcg related markov; do
  for o in "" "--opt fast_spec=0"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-cli-wall --no-matrix --no-secondary --data $d $o 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', '${o:-default (two lanes)}', j['value'], j['ms_per_step'])"
  done
done
for o in "" "--opt fast_spec=0"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-cli-wall --no-matrix --no-secondary --genomes 256 --length 100000 --rows-per-step 256 $o 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256x100k', '${o:-default (two lanes)}', j['value'], j['ms_per_step'])"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-cli-wall --no-matrix --no-secondary --rows-per-step 336 $o 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('336 rows', '${o:-default (two lanes)}', j['value'], j['ms_per_step'])"
done
