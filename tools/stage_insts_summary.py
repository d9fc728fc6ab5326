#!/usr/bin/env python3
"""Per-stage instruction counts of og_celt_kernel from the passes of tools/stage_insts.sh: python tools/stage_insts_summary.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
names = {1: "staging", 2: "level 0", 3: "fill levels", 5: "merges", 6: "anti-collapse", 7: "denormalise", 0: "IMDCT + row store"}
order = [1, 2, 3, 5, 6, 7, 0]
tot = {}
for k in order:
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{root}/stop{k}/**/*counter_collection.csv", recursive=True):
        per = defaultdict(float)
        kn = {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            kn[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (d, c), v in per.items():
            if kn[d].startswith("og_celt_kernel"):
                acc[c][0] += v
                acc[c][1] += 1
    tot[k] = {c: v[0] / v[1] for c, v in acc.items()}
frames = 32768
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
print(f"{'stage':20s}" + "".join(f"{c[3:]:>18s}" for c in cols) + "   (per frame: wave-instructions over the workgroup's 4 waves; LDS cycles)")
prev = {c: 0.0 for c in cols}
for k in order:
    row = tot[k]
    print(f"{names[k]:20s}" + "".join(f"{(row.get(c, 0) - prev[c]) / frames:18.0f}" for c in cols))
    prev = {c: row.get(c, 0) for c in cols}
print(f"{'total':20s}" + "".join(f"{prev[c] / frames:18.0f}" for c in cols))
