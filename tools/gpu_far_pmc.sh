#!/bin/bash
# Dev aid (GPU box): L1 / L2 / TA / SQ counters of ONE snk_fast_kernel launch with and without far chains, one rocprofv3
# --pmc pass per counter group (no trace domains).  Writes OUT.json: {config: {counter: value of the largest dispatch}}.
# Usage: tools/gpu_far_pmc.sh OUT.json COMMIT N L ROWS cfg...     cfg = FAR_LANESxFAR_WAVES (0x0 = LDS waves only)
OUT=${1:?usage}; COMMIT=${2:?COMMIT (tools/commit_id.sh)}; N=$3; L=$4; R=$5; shift 5
export TMPDIR=/tmp
D=$(dirname "$OUT")/farpmc_raw; mkdir -p "$D"
GROUPS_=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
         "TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
for cfg in "$@"; do
  fl=${cfg%x*}; fw=${cfg#*x}; g=0
  for grp in "${GROUPS_[@]}"; do
    timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d "$D/$cfg/$g" -- python3 tools/gpu_far_prof.py $N $L $R $fl $fw > "$D/$cfg.$g.log" 2>&1 \
      || { echo "group $g of $cfg failed:"; grep -m3 -i "error\|exceeds" "$D/$cfg.$g.log"; }
    echo "$cfg group $g: $(tail -1 "$D/$cfg.$g.log" | cut -c1-200)"
    g=$((g+1))
  done
done
python3 - "$D" "$OUT" "$COMMIT" $N $L $R "$@" <<'PY'
import csv, glob, json, sys, collections
d, out, commit, n, l, r = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
res = {"source": "rocprofv3 --pmc <group> (one pass per group) -- python3 tools/gpu_far_prof.py; largest snk_fast_kernel dispatch",
       "collected_at_commit": commit, "genomes": n, "length": l, "rows": r, "configs": {}}
for cfg in sys.argv[7:]:
    tot = {}
    for f in glob.glob(f"{d}/{cfg}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith("snk_fast_kernel"):
                per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
        if per:
            best = max(per.values(), key=lambda c: sum(c.values()))
            tot.update(best)
    logs = sorted(glob.glob(f"{d}/{cfg}.*.log"))
    tot["_run_lines"] = [open(x).read().strip().split("\n")[-1] for x in logs]
    res["configs"][cfg] = tot
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:3000])
PY
rm -rf "$D"
