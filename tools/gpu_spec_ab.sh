#!/bin/bash
for d in lcg related markov; do
  for o in "" "--opt fast_spec=1"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-cli-wall --no-matrix --data $d $o 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', '$o', j['value'], j['ms_per_step'])"
  done
done
for o in "" "--opt fast_spec=1"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-cli-wall --no-matrix --genomes 256 --length 100000 --rows-per-step 256 $o 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256x100k', '$o', j['value'], j['ms_per_step'])"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-cli-wall --no-matrix --rows-per-step 336 $o 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('336 rows', '$o', j['value'], j['ms_per_step'])"
done
