#!/bin/bash
# tools/profile_workloads.sh ROUND -- run ON THE GPU BOX (via gpurun) from the repo root: a round's rocprofv3 evidence (ROUND = r04, ...).
# For each single-GPU workload of bench.py (headline + the `configs` legs): one --kernel-trace --stats pass, then one --pmc pass
# per counter group (never combined with other tracing domains).  Summaries land in gpurun_out/<ROUND>_prof/<tag>/ ;
# tools/traffic_json.py turns them into profiles/<ROUND>_traffic.json (python3 tools/traffic_json.py ROUND).
# Counter passes serialise kernels, so nothing may wait for a kernel of another queue there.  The library notices the attached tool
# by itself (ssd_profiler_attached: ROCP_TOOL_LIBRARIES is set by rocprofv3) and uses host-side waits: NO SSD_AQL_SYNC is set here
# -- the counter passes are the check that the automatic choice works.  The kernel-trace pass runs with
# SSD_AQL_SYNC=0: tracing does not serialise kernels, and the trace should show the dispatch path the unprofiled run takes.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROUND=${1:?round tag, e.g. r04}
ROOT=gpurun_out/${ROUND}_prof
mkdir -p $ROOT
prof() {
  TAG=$1; PMCS=$2; shift; shift
  OUT=$ROOT/$TAG
  mkdir -p "$OUT"
  ARGS="--steps 300 --warmup 50 --no-extras $*"
  SSD_AQL_SYNC=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1
  cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
  grep '^{' "$OUT/trace.log" | tail -1 > "$OUT/bench_under_trace.json"
  echo "== $TAG: trace done"; head -3 "$OUT/kernel_stats.csv" | cut -c1-160
  i=0
  : > "$OUT/pmc_summary.txt"
  if [ "$PMCS" = "pmc" ]; then
  for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 bench.py $ARGS > "$OUT/pmc$i.log" 2>&1
    echo "   pmc pass $i rc=$?"
    F=$(find "$OUT/pmc$i" -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python3 tools/pmc_summary.py "$F" >> "$OUT/pmc_summary.txt"
    rm -rf "$OUT/pmc$i"
  done
  fi
  rm -rf "$OUT/trace"
}
prof harvest_16x38_n5_e4096 pmc
prof cleanup_25x18_n5_e4096 pmc --game cleanup
prof harvest_25x38_n5_e4096 pmc --game harvest25x38
prof cleanup_48x36_n10_e2048 pmc --game cleanup48x36 --envs 2048
prof harvest_16x38_n5_e4096_f32 pmc --obs-f32
echo "profile_workloads $ROUND done"
