#!/bin/bash
# tools/queue_probe_sweep.sh -- run on the GPU box: where is the hardware-queue cliff, and does the probe see it?
# The test-hook build lets the pool grow past the product's three queues (SSD_AQL_POOL_MAX); SSD_AQL_PROBE=0 keeps every queue.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c; mkdir -p $O
H=sequential_social_dilemma_games_amd/libssd_hip_testhooks.so
: > $O/queue_probe3.txt
for SIDE in 0 1 3; do
  for Q in 2 3 4 6; do
    echo "=== side $SIDE queues asked $Q probe on (test hooks)" >> $O/queue_probe3.txt
    SSD_LIB_PATH=$H SSD_AQL_POOL_MAX=8 SSD_AQL_VERBOSE=1 SSD_AQL_QUEUES=$Q timeout -k 5 120 python tools/queue_probe.py $SIDE 0 >> $O/queue_probe3.txt 2>&1
  done
done
grep -v "amdgpu.ids\|^#" $O/queue_probe3.txt | cut -c1-400
