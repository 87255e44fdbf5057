#!/bin/bash
# Dev aid (GPU box): SQ / LDS / TCP counters of one snk_fast_kernel launch, one rocprofv3 --pmc pass per group.
# Usage: [LANES=21 WAVES=4] tools/gpu_pmc.sh OUTDIR [N L ROWS]   (summaries: OUTDIR/pmc_<group>.json)
set -u
OUT=${1:-gpurun_out/pmc}; N=${2:-1024}; L=${3:-1000000}; R=${4:-84}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {   # group name, counters...
  g=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "snk_fast_" --output-format csv -d "$OUT/raw_$g" -- python3 tools/gpu_prof.py $N $L $R ${LANES:-21} ${WAVES:-4} > "$OUT/run_$g.log" 2>&1
  f=$(find "$OUT/raw_$g" -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$g" > "$OUT/pmc_$g.json" <<'PY'
import csv, sys, json, collections
tot = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("snk_fast_"):
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(json.dumps({"group": sys.argv[2], "kernel": "snk_fast_* (pair launches of the run: a 2-row warm-up + the measured one)", "per_launch": tot}))
PY
  cat "$OUT/pmc_$g.json"; tail -1 "$OUT/run_$g.log"; rm -rf "$OUT/raw_$g"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
