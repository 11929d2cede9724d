#!/bin/bash
# tools/profile_driver_cmd.sh ROUND -- run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the driver's exact bench command
# (python3 bench.py --gpus 1 --steps 20 --warmup 5); the summary lands in gpurun_out/${ROUND}_prof/driver_command/kernel_stats.csv.
set -u
ROUND=${1:?round tag, e.g. r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${ROUND}_prof/driver_command
mkdir -p "$OUT"
SSD_AQL_SYNC=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/trace.log" 2>&1
echo "rc=$?"
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
grep '^{' "$OUT/trace.log" | tail -1 > "$OUT/bench_under_trace.json"
rm -rf "$OUT/trace"
head -8 "$OUT/kernel_stats.csv" | cut -c1-200
