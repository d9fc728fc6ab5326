#!/usr/bin/env python3
"""profiles/pmc_traffic.json from one tools/gpu_measure.sh session: per workload (c2 c3 c4 c5) and kernel, per launch --
HBM bytes from the FETCH_SIZE / WRITE_SIZE passes, the VALU issue fraction from SQ_INSTS_VALU and GRBM_GUI_ACTIVE of the same
dispatches, and the kernel's average duration from the session's rocprofv3 --kernel-trace --stats run.
Units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both size counters are in KiB; FETCH_SIZE
reports half of the bytes of wide coalesced reads, so it is doubled (the kernels' bulk reads are 16 bytes per lane; their
dword-per-lane reads are uncalibrated, so the absolute figure is indicative); WRITE_SIZE is exact for 16-byte stores.
VALU issue fraction = 4 * SQ_INSTS_VALU / (1024 SIMDs * 2.4 GHz * the kernel's average duration in the --kernel-trace run): a
wave64 VALU instruction occupies its SIMD's issue port for four cycles, so 1.0 means every SIMD issued VALU work in every cycle
of the dispatch.  (GRBM_GUI_ACTIVE of the counter pass is kept beside it: dispatches run longer under the counter pass -- K1 by
70 % -- so cycles counted there would understate the fraction.)
usage: pmc_traffic.py gpurun_out/<tag> > profiles/pmc_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def counters(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{d}/pass*/**/*counter_collection.csv", recursive=True):
        per, names = defaultdict(float), {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])      # rows are per dimension instance (XCD / SE ...)
            names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        for (disp, c), v in per.items():
            a = acc[names[disp]][c]
            a[0] += v
            a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items()} | {"dispatches": max(v[1] for v in cs.values())} for k, cs in acc.items() if "og_" in k}


def durations(path):
    out = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            out[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e3
    return out


STEPS_RUN = 7      # tools/gpu_measure.sh runs the --pmc passes with --steps 5 --warmup 2 (and --no-verify): seven steps per process
doc = {"source": root, "note": __doc__.split("usage:")[0].strip(), "steps_per_pmc_run": STEPS_RUN}
for w in ("c2", "c3", "c4", "c5"):
    cs = counters(os.path.join(root, f"{w}_pmc"))
    if not cs:
        continue
    du = durations(os.path.join(root, f"{w}_kernel_stats.csv"))
    ks = {}
    for k, v in cs.items():
        e = {"avg_us": round(du[k], 2) if k in du else None, "dispatches_per_step": round(v["dispatches"] / STEPS_RUN, 2)}
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["hbm_bytes"] = int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
        if "SQ_INSTS_VALU" in v and k in du:
            e["insts_valu"] = int(v["SQ_INSTS_VALU"])
            e["insts_salu"] = int(v.get("SQ_INSTS_SALU", 0))
            e["insts_lds"] = int(v.get("SQ_INSTS_LDS", 0))
            e["waves"] = int(v.get("SQ_WAVES", 0))
            e["valu_issue_frac"] = round(4 * v["SQ_INSTS_VALU"] / (1024 * 2.4e9 * du[k] * 1e-6), 4)
            e["gui_active_cycles_per_xcd_in_pmc_pass"] = int(v.get("GRBM_GUI_ACTIVE", 0) / 8)
        ks[k] = e
    # per step: a kernel may be launched several times per step (c4: one SILK launch per depth), so means per dispatch x dispatches per step
    doc[w] = {"kernels": ks, "launch_bytes": int(sum(e.get("hbm_bytes", 0) * e["dispatches_per_step"] for e in ks.values())),
              "insts_valu_per_step": int(sum(e.get("insts_valu", 0) * e["dispatches_per_step"] for e in ks.values()))}
if "c2" in doc:      # (the keys bench.py read before round 3)
    doc["per_kernel_bytes"] = {k: e.get("hbm_bytes") for k, e in doc["c2"]["kernels"].items()}
    doc["launch_bytes"] = doc["c2"]["launch_bytes"]
    doc["og_celt_kernel_bytes"] = doc["c2"]["kernels"].get("og_celt_kernel", {}).get("hbm_bytes")
json.dump(doc, sys.stdout, indent=1)
print()
