#!/bin/bash
# K1 by size class (run on the GPU box from the repo root): bash tools/k1_classes.sh <outdir under gpurun_out>
# OPUSGPU_K1_CLASSES = bit mask of the classes og_pvq_all_kernel runs (1: N <= 176, 2: <= 64, 4: <= 24, 8: <= 8); the decoded
# audio is meaningless in the partial runs.  MASKS="15 8" limits the runs.
set -o pipefail
out="$PWD/gpurun_out/$1"
repo="$PWD"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for m in ${MASKS:-15 12 3 1 2 4 8}; do
  OPUSGPU_K1_CLASSES=$m timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/m$m" -- python3 "$repo/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg > "$out/m$m.log" 2>&1 || { echo "mask $m failed"; tail -5 "$out/m$m.log"; exit 1; }
  f=$(find "$out/m$m" -name '*kernel_stats.csv' | head -1)
  echo "mask $m: $(grep og_pvq_all_kernel "$f" | awk -F, '{print $(NF-4)}') ns avg"
done
