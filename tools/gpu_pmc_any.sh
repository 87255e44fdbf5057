#!/bin/bash
# Dev aid (GPU box): one rocprofv3 --pmc pass per counter group over one bench step of a codec, with a chosen build of the
# library; prints the sums over the largest dispatch of the named kernel.
# Usage: tools/gpu_pmc_any.sh OUTDIR CODEC KERNEL_SUBSTR LIB "CTR CTR ..." ["CTR CTR ..." ...]
OUT=${1:?}; CODEC=${2:?}; KSUB=${3:?}; LIB=${4:?}; shift 4
export TMPDIR=/tmp; mkdir -p "$OUT"
g=0
for grp in "$@"; do
  g=$((g+1)); D="$OUT/raw_$g"; rm -rf "$D"
  SNACC_HIP_LIB=$PWD/$LIB rocprofv3 --pmc $grp --output-format csv -d "$D" -- python3 bench.py --codec $CODEC --steps 1 --warmup 0 --no-cpu-baseline --no-matrix --no-cli-wall > "$OUT/run_$g.log" 2>&1 || { echo "group $g failed: $grp"; tail -3 "$OUT/run_$g.log"; continue; }
  f=$(find "$D" -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$KSUB" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
if per:
    best = max(per.values(), key=lambda d: sum(d.values()))
    print({k: v for k, v in sorted(best.items())})
PY
  rm -rf "$D"
done
