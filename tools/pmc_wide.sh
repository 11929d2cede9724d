#!/bin/bash
# tools/pmc_wide.sh TAG [bench args] -- GPU box: a WIDE counter sweep of one bench.py workload (one rocprofv3 --pmc pass per group,
# never combined with other tracing domains; kernels run one at a time under --pmc and the library waits on the host by itself).
# What the step kernel's waves wait for: instruction cache, scalar cache, the vector-memory address / data paths, L2, the fabric.
# Summaries: gpurun_out/pmc_wide/<TAG>.txt (tools/pmc_summary.py lines, env kernels only).
# Groups are sized to the blocks' counter slots per pass on gfx950 (SQ 8, TCC 4, TCP 4, TA 2, GRBM 2).  Round 3's sweep had TA / TCP /
# TCC groups of 5 - 8 counters: rocprofiler-sdk refuses such a group at the process's first HIP call ("Could not construct profile
# cfg ... error code 38: Request exceeds the capabilities of the hardware to collect"), aborts inside its tool library, and the
# process then sits in the tool's signal handler until the time bound kills it -- the four rc=124 passes of
# profiles/r03_dispatch/pmc_wide_harvest.txt were that (their logs end one second into the run, in ssd_create's first hipMalloc),
# not a wait of the library's.  A pass that is refused now says so and is cut short.
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROOT=gpurun_out/pmc_wide
mkdir -p $ROOT
python3 tools/_label.py "pmc_wide $TAG $*" > "$ROOT/$TAG.txt"
ARGS="--steps 200 --warmup 30 --no-extras $*"
i=0
while read -r PMC; do
  [ -z "$PMC" ] && continue
  i=$((i+1))
  OUT=$ROOT/${TAG}_p$i
  rm -rf "$OUT"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -- python3 bench.py $ARGS > "$OUT.log" 2>&1 &
  PID=$!
  # (a refused counter group leaves the process hanging in the profiler's abort handler: look at the log, do not wait 150 s for it)
  for t in $(seq 1 150); do
    sleep 1
    kill -0 $PID 2>/dev/null || break
    if grep -q "exceeds the capabilities of the hardware" "$OUT.log" 2>/dev/null; then sleep 2; kill -9 $PID 2>/dev/null; break; fi
  done
  wait $PID; rc=$?
  if grep -q "exceeds the capabilities of the hardware" "$OUT.log" 2>/dev/null; then echo "pass $i ($PMC): REFUSED by rocprofiler-sdk: more counters of one block than it has slots per pass" >> "$ROOT/$TAG.txt"; rm -rf "$OUT"; echo "pass $i refused"; continue; fi
  F=$(find "$OUT" -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$F" ]; then python3 tools/pmc_summary.py "$F" | grep "ssd_env_kernel" >> "$ROOT/$TAG.txt"; else echo "pass $i ($PMC): no counters, rc=$rc: $(tail -2 $OUT.log | tr '\n' ' ' | cut -c1-300)" >> "$ROOT/$TAG.txt"; fi
  rm -rf "$OUT"
  echo "pass $i rc=$rc"
done <<'EOF'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_IFETCH SQ_LEVEL_WAVES
SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL
TA_TA_BUSY_sum TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum
TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum
SPI_CSN_BUSY SPI_CSN_WAVE SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN GRBM_GUI_ACTIVE GRBM_CP_BUSY
EOF
echo "pmc_wide $TAG done"
