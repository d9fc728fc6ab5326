#!/bin/bash
# Run on the GPU box from the repo root: the SILK workload (bench.py --workload c3) and a kernel-trace profile of it.
#   bash tools/gpu_check_c3.sh <tag>      -> gpurun_out/<tag>/{c3.json,c3_kernel_stats.csv}
set -o pipefail
tag="${1:-check}"
R="$PWD"
out="$R/gpurun_out/$tag"
mkdir -p "$out"
timeout -k 10 300 python bench.py --workload c3 > "$out/c3.json" 2> "$out/c3.err" || { tail -20 "$out/c3.err"; exit 1; }
cat "$out/c3.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3prof" -- python3 "$R/bench.py" --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-verify > "$out/c3prof.log" 2>&1 || { tail -20 "$out/c3prof.log"; exit 1; }
find "$out/c3prof" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$out/c3_kernel_stats.csv"
head -4 "$out/c3_kernel_stats.csv" | cut -c1-200
