#!/bin/bash
# Dev aid (GPU box): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and the kernel time of one byte-kernel launch.
# Usage: tools/gpu_bytes_pmc.sh OUTDIR MODE N L ROWS [option=value ...]
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 tools/gpu_bytes_prof.py "$@" > "$OUT/$c.log" 2>&1
  echo "$c done" >> "$OUT/progress.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
d = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True)
    per = {}
    for row in csv.DictReader(open(f[0])):
        if row["Kernel_Name"].startswith("snk_bytes") and row["Counter_Name"] == c:
            k = (row["Kernel_Name"][:32], row["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
    for k, v in per.items():
        print(c, k, "KB", v, "-> bytes", v * 1024 * (2 if c == "FETCH_SIZE" else 1))
    print(open(f"{d}/{c}.log").read().strip().splitlines()[-1])
PY
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/gpu_bytes_prof.py "$@" > "$OUT/stats.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/stats/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if row["Name"].startswith("snk_bytes"):
        print("kernel_stats", row["Name"][:40], "calls", row["Calls"], "avg ns", row["AverageNs"], "total ns", row["TotalDurationNs"])
PY
grep "pair-compr" "$OUT/stats.log"
