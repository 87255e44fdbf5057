#!/bin/bash
# Dev aid (GPU box): every lz4 artefact of a round in one go -> OUT/ (copy what is to be judged into profiles/ as rNN_*):
#   bench.json (un-profiled bench line), kernel_stats_bench_*.csv (rocprofv3 --kernel-trace --stats of the same command),
#   pmc_traffic.json (HBM bytes, separate --pmc passes), pmc_sq.json (SQ / LDS / TCP counters + per-trip figures),
#   cycle_account.json (stats build), bench_related.json / bench_markov.json (secondary data sets).
# Usage: tools/gpu_round.sh OUT COMMIT        (COMMIT = tools/commit_id.sh in the build container)
OUT=${1:?usage: gpu_round.sh OUT COMMIT}; COMMIT=${2:?COMMIT required}
mkdir -p "$OUT"; export TMPDIR=/tmp
bash tools/gpu_profiles.sh "$OUT" "$COMMIT" lz4 || exit 1
if [ -f snacc_amd/libsnacc_hip_stats.so ]; then
  SNACC_HIP_LIB=$PWD/snacc_amd/libsnacc_hip_stats.so python3 tools/gpu_account.py 256 1000000 "$COMMIT" > "$OUT/cycle_account.json" 2> "$OUT/cycle_account.err"; echo "account done"
fi
bash tools/gpu_pmc.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1; echo "pmc groups: $(ls $OUT/pmc/pmc_*.json | wc -l)"
python3 tools/pmc_sq_summary.py "$OUT/pmc" "$COMMIT" "$OUT/bench.json" "$OUT/pmc_sq.json" "$OUT/cycle_account.json" || exit 1
for d in related markov; do
  python3 bench.py --data $d --cpu-seconds 5 > "$OUT/bench_$d.json" 2> "$OUT/bench_$d.err" || exit 1
  echo "bench $d: $(cut -c1-160 $OUT/bench_$d.json)"
done
