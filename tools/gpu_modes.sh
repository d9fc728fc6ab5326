#!/bin/bash
# Run on the GPU box from the repo root: bench.py --workload c3 / c4 under rocprofv3 --kernel-trace --stats.
#   bash tools/gpu_modes.sh <tag> [workloads...]   -> gpurun_out/<tag>/{c3,c4}.log + {c3,c4}_kernel_stats.csv
set -o pipefail
tag="${1:-modes}"; shift
R="$PWD"; out="$R/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for w in ${@:-c3 c4}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$w" -- python3 "$R/bench.py" --workload $w --steps 10 --warmup 2 --no-cpu-baseline > "$out/$w.log" 2>&1 || { tail -20 "$out/$w.log"; exit 1; }
  f=$(find "$out/prof_$w" -name '*kernel_stats.csv' | head -1); cp "$f" "$out/${w}_kernel_stats.csv"
  echo "$w: $(grep -o '"ms_per_step": [0-9.]*\|"verified": [a-z]*' "$out/$w.log" | tr '\n' ' ')"
  python3 "$R/tools/kstat_print.py" "$f"
done
