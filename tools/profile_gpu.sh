#!/bin/bash
# tools/profile_gpu.sh <tag> [bench args...] -- run ON THE GPU BOX (via gpurun) from the repo root.
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, each in its own run (never combined with
# tracing domains other than --kernel-trace).  Summaries land in gpurun_out/<tag>/ ; copy what you
# want judged into profiles/.
set -u
TAG=${1:-prof}; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 300 --warmup 50 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 bench.py $ARGS > "$OUT/pmc$i.log" 2>&1
  F=$(find "$OUT/pmc$i" -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 tools/pmc_summary.py "$F" >> "$OUT/pmc_summary.txt"
done
cat "$OUT/kernel_stats.csv"
cat "$OUT/pmc_summary.txt"
