#!/bin/bash
# Run on the GPU box from the repo root: step time of bench.py for --split 1, 2, 4.   bash tools/split_sweep.sh <tag>
set -o pipefail
out="gpurun_out/${1:-split}"
mkdir -p "$out"
for P in 1 2 4; do
  timeout -k 10 120 python bench.py --split $P --steps 60 --warmup 5 --no-cpu-baseline --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg > "$out/P$P.json" 2> "$out/P$P.err" || { tail -5 "$out/P$P.err"; exit 1; }
  python3 -c "import json; d=json.load(open('$out/P$P.json')); print('split', $P, d['ms_per_step'], d['verified'])"
done
