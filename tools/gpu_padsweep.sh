#!/bin/bash
# Dev aid (GPU box): trip-time sensitivity of the 2-bit steady loop to 4 padding instructions at points A..E
# (needs `make -C snacc_amd/csrc pads`).  Usage: tools/gpu_padsweep.sh [N L ROWS]
N=${1:-1024}; L=${2:-1000000}; R=${3:-84}
for p in "" padA padB padC padD padE; do
  lib=snacc_amd/libsnacc_hip${p:+_$p}.so
  echo -n "${p:-base} "; SNACC_HIP_LIB=$PWD/$lib python3 tools/gpu_sweep.py $N $L $R 21x4 2>/dev/null | tail -1
done
