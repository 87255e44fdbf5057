"""Dev aid: per-dispatch sum of one PMC counter for kernels whose name contains PATTERN.
Usage: pmc_sum.py counter_collection.csv PATTERN"""
import csv, sys, collections, json
per = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print(json.dumps({"counter_csv": sys.argv[1].split("/")[-1], "per_dispatch": per}))
