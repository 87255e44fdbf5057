#!/bin/bash
# Dev aid (GPU box): the exception sets of a round in one go (tools/gpu_exc.py 1024 1000000 84): the shipped library, the
# previous round's library beside it on the same box when snacc_amd/libsnacc_hip_PREV.so is there, and the stats build's
# account of the 5 % soft-masked set.  tools/exc_json.py turns the log into profiles/rNN_exceptions.json.
# Usage: tools/gpu_exc_round.sh OUT.log [PREV]          (PREV = r03 ...)
OUT=${1:?usage: gpu_exc_round.sh OUT.log [PREV]}; PREV=$2
KINDS="pure soft1 soft5 soft20 soft30 soft40 soft60 lower iupac20 iupac100 iupac1000 n10x100 n1x1000 n10x100+iupac20"
: > "$OUT"
echo "== shipped" >> "$OUT"; timeout -k 10 400 python3 tools/gpu_exc.py 1024 1000000 84 $KINDS 2>/dev/null | grep -v stats >> "$OUT" || exit 1
if [ -n "$PREV" ] && [ -f snacc_amd/libsnacc_hip_$PREV.so ]; then
  echo "== $PREV" >> "$OUT"; SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_$PREV.so timeout -k 10 600 python3 tools/gpu_exc.py 1024 1000000 84 $KINDS 2>/dev/null | grep -v stats >> "$OUT" || exit 1
fi
if [ -f snacc_amd/libsnacc_hip_stats.so ]; then
  echo "== stats build" >> "$OUT"; SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_stats.so timeout -k 10 300 python3 tools/gpu_exc.py 256 1000000 84 soft5 soft20 2>/dev/null >> "$OUT" || exit 1
fi
echo "exceptions done"
