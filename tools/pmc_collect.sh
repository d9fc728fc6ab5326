#!/bin/bash
# One rocprofv3 --pmc pass per counter group over a short bench.py run (run on the GPU box from the repo root):
#   bash tools/pmc_collect.sh <outdir under gpurun_out> [bench args...]
# Each pass is its own process under its own time limit; a failing pass ends the script.
set -o pipefail
out="gpurun_out/$1"; shift
args=("$@")
[ ${#args[@]} -eq 0 ] && args=(--steps 5 --warmup 2 --no-cpu-baseline --no-t1 --no-overlap-leg --no-e2e --no-pipeline-leg)
repo="$PWD"
mkdir -p "$repo/$out"
cd /tmp && export TMPDIR=/tmp
groups=(
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  i=$((i + 1))
  timeout -s KILL 150 rocprofv3 --pmc $g --output-format csv -d "$repo/$out/pass$i" -- python3 "$repo/bench.py" "${args[@]}" > "$repo/$out/pass$i.log" 2>&1 || { echo "pass $i ($g) failed"; tail -5 "$repo/$out/pass$i.log"; exit 1; }
  echo "pass $i ok: $g"
done
