#!/bin/bash
# One measurement session on the GPU box (run from the repo root): for each workload (c2 c3 c4 c5) the bench line, a
# rocprofv3 --kernel-trace --stats run and three --pmc passes (instruction mix + busy cycles, FETCH_SIZE, WRITE_SIZE: each pass a
# process of its own under its own time limit), all from the same build.  tools/pmc_traffic.py turns the directory into
# profiles/pmc_traffic.json (HBM bytes and VALU issue fraction per kernel).
#   bash tools/gpu_measure.sh <tag> [workloads...]   -> gpurun_out/<tag>/...
set -o pipefail
tag="${1:-measure}"; shift
R="$PWD"; out="$R/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
groups=(
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
for w in ${@:-c2 c3 c4 c5}; do
  if [ "$w" = c2 ]; then args=(--no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg); else args=(--workload "$w"); fi
  timeout -k 10 240 python3 "$R/bench.py" "${args[@]}" --no-cpu-baseline > "$out/${w}_bench.json" 2> "$out/${w}_bench.err" || { tail -20 "$out/${w}_bench.err"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${w}_prof" -- python3 "$R/bench.py" "${args[@]}" --steps 10 --warmup 2 --no-cpu-baseline > "$out/${w}_prof.log" 2>&1 || { tail -20 "$out/${w}_prof.log"; exit 1; }
  f=$(find "$out/${w}_prof" -name '*kernel_stats.csv' | head -1); cp "$f" "$out/${w}_kernel_stats.csv"
  i=0
  for g in "${groups[@]}"; do
    i=$((i + 1))
    timeout -s KILL 150 rocprofv3 --pmc $g --output-format csv -d "$out/${w}_pmc/pass$i" -- python3 "$R/bench.py" "${args[@]}" --steps 5 --warmup 2 --no-cpu-baseline --no-verify > "$out/${w}_pmc_pass$i.log" 2>&1 || { echo "$w pass $i ($g) failed"; tail -5 "$out/${w}_pmc_pass$i.log"; exit 1; }
  done
  echo "$w: $(grep -o '"ms_per_step": [0-9.]*' "$out/${w}_bench.json" | head -1)"
  python3 "$R/tools/kstat_print.py" "$out/${w}_kernel_stats.csv"
done
python3 "$R/tools/pmc_traffic.py" "$out" > "$out/pmc_traffic.json" && cat "$out/pmc_traffic.json" | head -80
