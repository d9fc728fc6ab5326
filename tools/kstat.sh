#!/bin/bash
# Run on the GPU box from the repo root: kernel statistics of a short bench.py run only (no tests).  bash tools/kstat.sh <tag> [bench args]
set -o pipefail
tag="${1:-kstat}"; shift
R="$PWD"; out="$R/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg "$@" > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
find "$out/prof" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
grep '"verified"' "$out/prof.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'verified', d['verified'])"
python3 -c "import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if \"og_\" in r[\"Name\"][:12]: print(r[\"Name\"][:44], r[\"Calls\"], round(float(r[\"AverageNs\"])/1e3, 1))" "$out/kernel_stats.csv"
