#!/usr/bin/env python3
"""Aggregates the counter_collection.csv files of tools/pmc_collect.sh into per-kernel means per launch (JSON on stdout)."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])          # rows are per dimension instance (XCD / SE ...)
        names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    for (d, c), v in per_dispatch.items():
        a = acc[names[d]][c]
        a[0] += v
        a[1] += 1
out = {k: {c: round(v[0] / v[1], 1) for c, v in sorted(cs.items())} | {"launches_seen": max(v[1] for v in cs.values())}
       for k, cs in acc.items() if k.startswith("og_") or "og_" in k}
json.dump(out, sys.stdout, indent=1)
print()
