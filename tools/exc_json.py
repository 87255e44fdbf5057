"""Turns the log of tools/gpu_exc_round.sh into the round's exceptions summary.  Usage: exc_json.py LOG COMMIT OUT.json"""
import json
import re
import sys

log, commit, out = sys.argv[1:4]
res = {"what": "2-bit kernel on genomes with exceptions (tools/gpu_exc_round.sh = tools/gpu_exc.py 1024 1000000 84 <kind>, one row per chain of a CU: "
               "84, or 83 for a set with other-case stretches): pair-compressions per second against the pure-ACGT rate of the same run; "
               "soft<P> = P % lower case in stretches of ~500 bases, iupac<K> = K single IUPAC codes per sequence, n10x100 = ten runs of 100 N",
       "collected_at_commit": commit}
sec = None
for ln in open(log):
    ln = ln.rstrip("\n")
    if ln.startswith("== "):
        sec = {"shipped": "round4_kernel", "stats build": "stats_build_256_genomes"}.get(ln[3:], "previous_round_library_same_box_" + ln[3:])
        res[sec] = {} if sec != "stats_build_256_genomes" else {"lines": []}
        continue
    m = re.match(r"\s*(\S+)\s+rows=(\d+) packed=(\d+)/(\d+) ms=([\d.]+) pairs/s=(\d+) \((\d+)% of pure\) parity=(\w+)", ln)
    if sec == "stats_build_256_genomes":
        if ln.strip():
            res[sec]["lines"].append(ln.strip())
        continue
    if m and sec:
        res[sec][m.group(1)] = {"rows": int(m.group(2)), "packed": int(m.group(3)), "kernel_ms": float(m.group(5)),
                                "pair_compr_per_s": int(m.group(6)), "pct_of_pure": int(m.group(7)), "oracle_spot_check": m.group(8) == "True"}
json.dump(res, open(out, "w"), indent=1)
print(out, {k: len(v) for k, v in res.items() if isinstance(v, dict)})
