#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03e; mkdir -p $O
H=sequential_social_dilemma_games_amd/libssd_hip_testhooks.so
for KIND in 0 1; do for K in 1 20; do
  SSD_LIB_PATH=$H SSD_AQL_FORK_KIND=$KIND timeout -k 5 120 python tools/fork_ab.py $K 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/fork_ab.txt
done; done
timeout -k 5 120 python tools/fork_ab.py 20 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/fork_ab.txt
bash tools/matrix_c48.sh $O
