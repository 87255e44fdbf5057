#!/bin/bash
# Dev aid (GPU box): the profile artefacts of a round in one go -> OUT/ (copy what is to be judged into profiles/).
#   bench lines (un-profiled runs, with the CPU baselines), rocprofv3 --kernel-trace --stats of the same command,
#   HBM traffic (separate --pmc passes) for lz4 / gzip / zlib.
# COMMIT = what `tools/commit_id.sh` prints in the build container (the GPU box has no .git): it is written into every
# summary as `collected_at_commit`.
# Usage: tools/gpu_profiles.sh OUT COMMIT [codecs...]   (default: lz4 gzip zlib)
OUT=${1:?usage: gpu_profiles.sh OUT COMMIT [codecs...]}; COMMIT=${2:?COMMIT (tools/commit_id.sh) is required}; shift 2
CODECS=${@:-lz4 gzip zlib}
mkdir -p "$OUT"; export TMPDIR=/tmp
for c in $CODECS; do
  sfx=$([ $c = lz4 ] && echo "" || echo "_$c")
  python3 bench.py --codec $c > "$OUT/bench$sfx.json" 2> "$OUT/bench$sfx.err" || exit 1
  echo "bench $c: $(cut -c1-160 $OUT/bench$sfx.json)"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$c" -- python3 bench.py --codec $c --steps 4 --warmup 1 --no-cpu-baseline --no-matrix --no-cli-wall --no-secondary > "$OUT/kt_$c.log" 2>&1 || exit 1
  f=$(find "$OUT/kt_$c" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$OUT/kernel_stats_bench${sfx}_1024x1Mbp_84rows.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        w.writerow([r[0][:140]] + r[1:])          # (kernel names cut to 140 characters)
PY
  rm -rf "$OUT/kt_$c"
  if [ $c = lz4 ]; then bash tools/gpu_traffic.sh "$OUT/pmc_traffic.json" "$COMMIT" > /dev/null
  else bash tools/gpu_traffic_dfl.sh "$OUT/pmc_traffic_$c.json" "$COMMIT" $c > /dev/null; fi
  echo "done $c"
done
