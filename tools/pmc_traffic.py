#!/usr/bin/env python3
"""HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_collect.sh (summary.json of tools/pmc_summary.py).
Units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB; FETCH_SIZE
reports half of the bytes of wide coalesced reads, so it is doubled (the kernels' bulk reads are 16 bytes per lane; their
dword-per-lane reads are uncalibrated, so the absolute figure is indicative); WRITE_SIZE is exact for 16-byte stores.
usage: pmc_traffic.py gpurun_out/<dir>/summary.json > profiles/pmc_traffic.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
per = {}
for k, v in d.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        per[k.replace("void ", "")] = int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
out = {"source": sys.argv[1], "unit": "bytes per launch (32768 frames: 4096 streams x 8 frames)",
       "per_kernel_bytes": per, "launch_bytes": sum(per.values()),
       "og_celt_kernel_bytes": per.get("og_celt_kernel")}
json.dump(out, sys.stdout, indent=1)
print()
