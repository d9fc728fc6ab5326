#!/bin/bash
# Run on the GPU box from the repo root: step time of bench.py with either form of the band-plan kernel (OPUSGPU_PLAN), for
# several launch sizes (frames per stream per step x 4096 streams).   bash tools/plan_forms.sh <tag>
set -o pipefail
out="gpurun_out/${1:-planforms}"
mkdir -p "$out"
for T in ${TS:-1 2 4 8}; do
  for form in lane steps; do
    OPUSGPU_PLAN=$form timeout -k 10 120 python bench.py --frames-per-step $T --steps 40 --warmup 4 --no-cpu-baseline --no-verify --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg > "$out/T${T}_$form.json" 2> "$out/T${T}_$form.err" || { tail -5 "$out/T${T}_$form.err"; exit 1; }
    python3 -c "import json,sys; d=json.load(open('$out/T${T}_$form.json')); print('T=$T', '$form', d['ms_per_step'])"
  done
done
