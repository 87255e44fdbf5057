#!/bin/bash
# Dev aid (GPU box): HBM traffic of one snk_fast_kernel launch of the bench shape, as the microarch guide prescribes
# (separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB units, FETCH_SIZE doubled on gfx950).  Writes OUT.json.
# Usage: [DATA=lcg|markov|related|softmask5] [KERNEL=snk_fast_kernel] tools/gpu_traffic.sh OUT.json COMMIT [N L ROWS]
OUT=${1:-gpurun_out/r02_pmc_traffic.json}; COMMIT=${2:?COMMIT (tools/commit_id.sh, run in the build container) is required}; N=${3:-1024}; L=${4:-1000000}; R=${5:-84}
export TMPDIR=/tmp
D=$(dirname "$OUT")/traffic_raw; mkdir -p "$D"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "snk_" --output-format csv -d "$D/$c" -- python3 tools/gpu_prof.py $N $L $R 21 4 > "$D/$c.log" 2>&1
done
python3 - "$D" "$OUT" "$COMMIT" $N $L $R "${DATA:-lcg}" "${KERNEL:-snk_fast_kernel}" <<'PY'
import csv, glob, json, sys
d, out, commit, n, l, r = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
data, kern = sys.argv[7], sys.argv[8]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith(kern) and row["Counter_Name"] == c:
            per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    vals = sorted(per.values())
    tot[c] = vals[len(vals) // 2] if vals else None            # median launch (the tool launches a 2-row warm-up first: take the big one)
    tot[c + "_per_dispatch_KB"] = per
fetch_kb = max(tot["FETCH_SIZE_per_dispatch_KB"].values()); write_kb = max(tot["WRITE_SIZE_per_dispatch_KB"].values())
res = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/gpu_prof.py, largest {kern} dispatch",
       "collected_at_commit": commit, "genomes": n, "length": l, "rows": r, "data": data, "kernel": kern,
       "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
       "note": "FETCH_SIZE doubled (gfx950: counts 128-B requests at 64 B); algorithmic bytes of the launch = rows*genomes*(2*length+4)",
       "algorithmic_bytes_per_launch": r * n * (2 * l + 4)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
PY
rm -rf "$D"
