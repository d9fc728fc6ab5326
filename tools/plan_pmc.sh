#!/bin/bash
# Run on the GPU box from the repo root: instruction counters of the band-plan kernel in the form OPUSGPU_PLAN names.
#   OPUSGPU_PLAN=steps bash tools/plan_pmc.sh <tag>
set -o pipefail
R="$PWD"; out="$R/gpurun_out/${1:-planpmc}"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d "$out/p1" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg > "$out/p1.log" 2>&1 || { tail -5 "$out/p1.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for f in glob.glob(sys.argv[1] + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    if "og_" in k: print(k, len(n[k]), {c: round(v / len(n[k]) / 1e6, 2) for c, v in acc[k].items()}, "(millions per launch)")
PY
