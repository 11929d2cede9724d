#!/bin/bash
# tools/pmc_once.sh "<COUNTERS...>" [bench args] -- one rocprofv3 --pmc pass over bench.py, per-kernel averages (GPU box).
set -u
PMC="$1"; shift
OUT=gpurun_out/pmc_once_$$
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline "$@" > "$OUT.log" 2>&1
python3 tools/pmc_summary.py $(find "$OUT" -name "*counter_collection.csv" | head -1) | grep ", 0>"
