#!/bin/bash
# Dev aid (GPU box): HBM traffic of one gzip / zlib bench step (86 016 pair jobs of dfl_parse_kernel*), as the microarch
# guide prescribes (separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB units, FETCH_SIZE doubled on gfx950).
# Usage: tools/gpu_traffic_dfl.sh OUT.json COMMIT [gzip|zlib]
OUT=${1:-gpurun_out/pmc_traffic_gzip.json}; COMMIT=${2:?COMMIT (tools/commit_id.sh, run in the build container) is required}; CODEC=${3:-gzip}
export TMPDIR=/tmp
D=$(dirname "$OUT")/traffic_raw_dfl; mkdir -p "$D"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$D/$c" -- python3 bench.py --codec $CODEC --steps 1 --warmup 0 --no-cpu-baseline --no-matrix --no-cli-wall > "$D/$c.log" 2>&1
done
python3 - "$D" "$OUT" "$COMMIT" "$CODEC" <<'PY'
import csv, glob, json, sys
d, out, commit, codec = sys.argv[1:5]
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --codec %s --steps 1 --warmup 0; "
                 "largest dfl_parse_kernel* dispatch = the step of 84 x 1024 pair jobs" % codec,
       "collected_at_commit": commit, "codec": codec, "genomes": 1024, "length": 1000000, "rows": 84}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for row in csv.DictReader(open(f)):
        if "dfl_parse_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
            per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    res[c + "_KB"] = max(per.values())
pairs = 84 * 1024
res["hbm_bytes_per_launch"] = (2.0 * res["FETCH_SIZE_KB"] + res["WRITE_SIZE_KB"]) * 1024.0
res["fetched_bytes_per_pair_x2"] = 2.0 * res["FETCH_SIZE_KB"] * 1024.0 / pairs
res["fetched_bytes_per_pair_raw"] = res["FETCH_SIZE_KB"] * 1024.0 / pairs
res["written_bytes_per_pair"] = res["WRITE_SIZE_KB"] * 1024.0 / pairs
res["algorithmic_bytes_per_pair"] = 2000004
res["note"] = "FETCH_SIZE doubled per the gfx950 calibration (an upper bound here: the reads are 4-8 B per lane)"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
PY
rm -rf "$D"
