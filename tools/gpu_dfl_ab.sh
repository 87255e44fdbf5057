#!/bin/bash
# Dev aid (GPU box): A/B of two builds of the library on the gzip / zlib bench step (same box, same data).
# Usage: tools/gpu_dfl_ab.sh OUT LIB_A LIB_B [bench options...]
OUT=${1:?usage: gpu_dfl_ab.sh OUT LIB_A LIB_B}; A=${2:?}; B=${3:?}; shift 3
mkdir -p "$OUT"
for codec in gzip zlib; do
  for lib in "$A" "$B"; do
    tag=$(basename "$lib" .so)
    SNACC_HIP_LIB=$PWD/$lib python3 bench.py --codec $codec --no-cpu-baseline --no-matrix --no-cli-wall "$@" > "$OUT/${codec}_$tag.json" 2> "$OUT/${codec}_$tag.err" || exit 1
    python3 -c "import json,sys; d=json.load(open('$OUT/${codec}_$tag.json')); print('$codec $tag', round(2*d['value']), 'pair-compr/s', round(d['ms_per_step'],1), 'ms')"
  done
done
