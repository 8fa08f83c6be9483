"""Aggregate rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch for kernels matching a substring.
usage: pmc_collect.py <out_dir> <kernel substring> [more dirs...] -> JSON on stdout"""
import csv, glob, json, sys
from collections import defaultdict
sub = sys.argv[2]
res = defaultdict(lambda: [0.0, 0])
for d in [sys.argv[1]] + sys.argv[3:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                k = r["Counter_Name"]
                res[k][0] += float(r["Counter_Value"])
                res[k][1] += 1
print(json.dumps({k: {"mean_per_dispatch": v[0] / max(v[1], 1), "dispatches": v[1]} for k, v in sorted(res.items())}, indent=1))
