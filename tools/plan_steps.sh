#!/bin/bash
# Run on the GPU box from the repo root: time of og_plan_kernel when every workgroup leaves after step A / B / not at all.
#   bash tools/plan_steps.sh <tag> [bench args]     (results of the stopped runs are garbage: --no-verify)
set -o pipefail
tag="${1:-plansteps}"; shift
R="$PWD"; out="$R/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for k in 3 1 2 0; do
  OPUSGPU_PLAN_STOP=$k OPUSGPU_STOP_AFTER=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/s$k" -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg "$@" > "$out/s$k.log" 2>&1 || { tail -5 "$out/s$k.log"; exit 1; }
  f=$(find "$out/s$k" -name '*kernel_stats.csv' | head -1)
  python3 -c "import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith('og_plan'): print('stop', $k, round(float(r['AverageNs'])/1e3, 1), 'us')" "$f"
done
