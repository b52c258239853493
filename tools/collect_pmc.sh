#!/bin/bash
# Collect the rocprofv3 PMC counters behind bench.py's `roofline.traffic` (run on the GPU box).
# One counter group per run, PMC only (no trace domains), as MI355X_MICROARCH.md prescribes:
#   FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) cannot share a pass; SQ counters go in a third.
# Usage: bash tools/collect_pmc.sh <tag>      -> gpurun_out/pmc_<tag>_summary.json
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-settle --no-configs \
    > "$OUT/$name.log" 2>&1
done
# the snapshot on the box has no .git: the caller passes the commit (ALOHA_COMMIT=$(git rev-parse --short HEAD) in front of gpurun's command)
cd "$ROOT" && python3 tools/pmc_summary.py "$OUT" > "$ROOT/gpurun_out/pmc_${TAG}_summary.json"
cat "$ROOT/gpurun_out/pmc_${TAG}_summary.json"
