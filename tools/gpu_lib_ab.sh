#!/bin/bash
# Dev aid (GPU box): A/B of two builds of the library on the lz4 bench step, interleaved (A B A B) on one box.
# Usage: tools/gpu_lib_ab.sh LIB_A LIB_B [bench options...]
A=${1:?usage: gpu_lib_ab.sh LIB_A LIB_B [bench options]}; B=${2:?}; shift 2
for rep in 1 2; do
  for lib in "$A" "$B"; do
    SNACC_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-matrix --no-cli-wall --steps 6 --warmup 2 "$@" 2>/dev/null | \
      python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $lib .so)', round(2*j['value']), 'pair-compr/s', round(j['ms_per_step'],2), 'ms; parity', j.get('parity'))"
  done
done
