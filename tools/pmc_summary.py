#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (csv) into one json: per counter, the average over launches of the named kernel
of the per-launch sum.  usage: pmc_summary.py <kernel substring> <out.json> <pass dir> [<pass dir> ...]"""
import csv, glob, json, os, sys
from collections import defaultdict

kern, out = sys.argv[1], sys.argv[2]
res = {}
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(f)):
            if kern not in row.get("Kernel_Name", ""):
                continue
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for c, by in per.items():
            res[c] = sum(by.values()) / len(by)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
