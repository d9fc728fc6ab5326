#!/usr/bin/env python3
"""Prints the og_* kernels of a rocprofv3 --stats kernel_stats.csv: name, calls, average and maximum duration in us."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].replace("void ", "")
    if name.startswith("og_"):
        print(f"{name[:36]:36s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:9.1f} {float(r['MaxNs']) / 1e3:9.1f}")
