export TMPDIR=/tmp
mkdir -p gpurun_out/r2z/pmcd
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r2z/pmcd/raw -- python3 bench.py --codec gzip --steps 1 --warmup 0 --no-cpu-baseline --no-matrix > gpurun_out/r2z/pmcd/run.log 2>&1
f=$(find gpurun_out/r2z/pmcd/raw -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "dfl_parse_kernel" in r["Kernel_Name"]:
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
best = max(per.values(), key=lambda d: d.get("SQ_WAVE_CYCLES", 0))
print({k: v for k, v in best.items()})
PY
rm -rf gpurun_out/r2z/pmcd/raw
