"""Dev aid: merge the per-group summaries of tools/gpu_pmc.sh (OUTDIR/pmc_<group>.json) into one record with the per-wave-trip
figures DESIGN.md section 6 quotes.  Usage: pmc_sq_summary.py OUTDIR COMMIT BENCH.json OUT.json [CYCLE_ACCOUNT.json]
BENCH.json: a bench.py line of the same launch shape (its issue_bound.probes_per_pair gives the probes of the launch);
CYCLE_ACCOUNT.json (tools/gpu_account.py): probes per chain-trip of the two-lane loop (1 if absent: one lane per chain)."""
import glob
import json
import sys

outdir, commit, bench, out = sys.argv[1:5]
ppt = json.load(open(sys.argv[5])).get("probes_per_chain_trip", 1.0) if len(sys.argv) > 5 else 1.0
line = json.load(open(bench))
rows, n = line["config"]["rows_per_step_per_gpu"], line["config"]["genomes"]
probes = line["issue_bound"]["probes_per_pair"]
lanes = 21
rec = {"source": "rocprofv3 --pmc (one pass per group, tools/gpu_pmc.sh), tools/gpu_prof.py %d x %d bp, %d rows, per launch of snk_fast_kernel "
                 "(persistent: 256 workgroups x 4 waves)" % (n, line["config"]["length"], rows),
       "collected_at_commit": commit}
for f in sorted(glob.glob(outdir + "/pmc_*.json")):
    rec.update(json.load(open(f))["per_launch"])
trips = rows * n * probes / lanes / ppt
q = 4.0                                            # SQ_*_CYCLES count quad-cycles
waves = rec["SQ_WAVES"]
rec["derived_per_wave_trip"] = {
    "trips": trips, "probes_per_chain_trip": ppt, "cycles": rec["SQ_WAVE_CYCLES"] * q / trips,
    "valu": rec["SQ_INSTS_VALU"] / trips, "salu": rec["SQ_INSTS_SALU"] / trips, "lds": rec["SQ_INSTS_LDS"] / trips,
    "vmem": (rec["SQ_INSTS_VMEM_RD"] + rec["SQ_INSTS_VMEM_WR"]) / trips,
    "busy_cycles": rec["SQ_ACTIVE_INST_ANY"] * q / trips if "SQ_ACTIVE_INST_ANY" in rec else None,
    "wait_cycles": rec["SQ_WAIT_ANY"] * q / trips if "SQ_WAIT_ANY" in rec else None,
    "lds_array_cycles": rec.get("SQ_LDS_IDX_ACTIVE", 0) / trips, "lds_bank_conflict_cycles": rec.get("SQ_LDS_BANK_CONFLICT", 0) / trips,
    "l1_accesses": rec.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / trips,
    "l1_hit_rate": 1.0 - rec.get("TCP_TCC_READ_REQ_sum", 0) / max(rec.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1), 1),
    "note": "SQ_*_CYCLES count quad-cycles (x4); a trip = one probe of role 0 (plus role 1's when it counts: probes_per_chain_trip) for each of the "
            "21 chains of a wave; trips = rows * genomes * probes_per_pair (oracle statistics, bench line) / 21 / probes_per_chain_trip"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec["derived_per_wave_trip"]))
