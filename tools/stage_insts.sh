#!/bin/bash
# Instruction counts of og_celt_kernel stage by stage (run on the GPU box from the repo root):
#   bash tools/stage_insts.sh <outdir under gpurun_out>
# OPUSGPU_STOP_AFTER=k makes every frame leave the kernel after stage k (1 staging, 2 band walk level 0, 3 fill levels,
# 5 merges, 6 anti-collapse, 7 denormalise, 0 everything); one rocprofv3 --pmc pass per value, differences = per stage.
# The decoded audio is meaningless in these runs.
set -o pipefail
out="$PWD/gpurun_out/$1"
repo="$PWD"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 5 6 7 0; do
  OPUSGPU_STOP_AFTER=$k timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/stop$k" -- python3 "$repo/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg > "$out/stop$k.log" 2>&1 || { echo "stop $k failed"; tail -5 "$out/stop$k.log"; exit 1; }
  echo "stop $k ok"
done
