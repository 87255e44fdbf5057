#!/bin/bash
# Dev aid (GPU box): SQ instruction mix of one gzip / zlib bench step (the largest dfl_parse_kernel* dispatch), one
# rocprofv3 --pmc pass.  Writes OUT.json.
# Usage: tools/gpu_pmc_dfl.sh OUT.json COMMIT [gzip|zlib]
OUT=${1:?usage: gpu_pmc_dfl.sh OUT.json COMMIT [gzip|zlib]}; COMMIT=${2:?COMMIT (tools/commit_id.sh) is required}; CODEC=${3:-gzip}
export TMPDIR=/tmp
D=$(dirname "$OUT")/pmcd_raw; mkdir -p "$D"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$D/raw" -- python3 bench.py --codec $CODEC --steps 1 --warmup 0 --no-cpu-baseline --no-matrix --no-cli-wall > "$D/run.log" 2>&1
f=$(find "$D/raw" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$OUT" "$COMMIT" "$CODEC" <<'PY'
import csv, sys, collections, json
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "dfl_parse_kernel" in r["Kernel_Name"]:
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
best = max(per.values(), key=lambda d: d.get("SQ_WAVE_CYCLES", 0))
res = {"source": "rocprofv3 --pmc (one pass) -- python3 bench.py --codec %s --steps 1 --warmup 0; largest dfl_parse_kernel* dispatch" % sys.argv[4],
       "collected_at_commit": sys.argv[3], "codec": sys.argv[4], "per_launch": dict(best)}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res))
PY
rm -rf "$D"
