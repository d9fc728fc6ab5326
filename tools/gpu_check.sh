#!/bin/bash
# Run on the GPU box from the repo root: GPU tests, the default bench line and a kernel-trace profile of the same command.
#   bash tools/gpu_check.sh <tag>      -> gpurun_out/<tag>/{pytest.log,bench.json,prof/...}
set -o pipefail
tag="${1:-check}"
R="$PWD"
out="$R/gpurun_out/$tag"
mkdir -p "$out"
timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
timeout -k 10 240 python bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
cat "$out/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
find "$out/prof" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
head -8 "$out/kernel_stats.csv" | cut -c1-200
