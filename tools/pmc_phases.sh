#!/bin/bash
# tools/pmc_phases.sh [harvest|cleanup] -- dynamic per-wave instruction counts of the step kernel with each phase
# skipped in turn (GPU box, diagnostic library).  Differences against the first line = that phase's instructions.
set -u
GAME=${1:-harvest}
export SSD_LIB_PATH=$GRAFT_REPO_ROOT/sequential_social_dilemma_games_amd/libssd_hip_stamps.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for CFG in "0 1 full" "0 0 no-obs" "1 1 no-move" "2 1 no-consume" "4 1 no-beams" "8 1 no-respawn" "15 0 floor"; do
  set -- $CFG
  OUT=gpurun_out/pmc_phase_$3
  rm -rf "$OUT"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d "$OUT" -- python3 tools/pmc_phases.py $1 $2 $GAME > "$OUT.log" 2>&1
  echo "== $3"
  python3 tools/pmc_summary.py $(find "$OUT" -name "*counter_collection.csv" | head -1) | grep "true, true" | awk '{print "   ", $(NF-3), $(NF-1)/4096}'
done
