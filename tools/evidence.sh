#!/bin/bash
# tools/evidence.sh ROUND [1|2] -- run ON THE GPU BOX (part 1: fuzz + the shipped maps' soaks; part 2: the enlarged maps and the large output rings; none: both): the long parity runs of the final build whose logs are kept under profiles/:
# the fuzz tool (400 random configurations through every API path, ssd_rollout_actions with action / order rings included) and the
# soak (4096 envs x 5000 steps per game as rollout chains -- the library's own dispatch queues, coherent, split --, the same with
# caller-supplied actions (ssd_rollout_actions: chains and fused), and with the test-hook library's SSD_AQL_ALTERNATE=1, which
# moves every env to another workgroup / XCD from one launch to the next).  Every log starts with the library that ran.
ROUND=${1:?round tag, e.g. r04}
cd $GRAFT_REPO_ROOT
D=gpurun_out/${ROUND}_evidence
mkdir -p $D
PART=${2:-0}
if [ "$PART" != "2" ]; then
python3 tools/fuzz_parity.py 400 0 2>&1 | grep -v amdgpu.ids | tee $D/fuzz_400_seed0.log | tail -3
FUZZ_BIG=1 python3 tools/fuzz_parity.py 100 7 2>&1 | grep -v amdgpu.ids | tee $D/fuzz_big_100_seed7.log | tail -2
for g in harvest cleanup; do
  python3 tools/soak_parity.py $g 4096 5000 250 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_${g}_chains.log | tail -2
  python3 tools/soak_parity.py $g 4096 3000 250 actions 2>&1 | grep -v amdgpu.ids | tee $D/soak_${g}_actions.log | tail -2
  python3 tools/soak_parity.py $g 4096 2000 250 actions_fused 2>&1 | grep -v amdgpu.ids | tee $D/soak_${g}_actions_fused.log | tail -2
  SSD_LIB_PATH=sequential_social_dilemma_games_amd/libssd_hip_testhooks.so SSD_AQL_ALTERNATE=1 SSD_AQL_ALWAYS_FORK=1 python3 tools/soak_parity.py $g 4096 3000 250 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_${g}_chains_alternate.log | tail -2
  python3 tools/soak_parity.py $g 4096 3000 250 fused 2>&1 | grep -v amdgpu.ids | tee $D/soak_${g}_fused.log | tail -2
done
fi
if [ "$PART" = "1" ]; then exit 0; fi
# ... and the enlarged maps' own kernels (BASELINE.json's 25x38 label; configs[4]'s per-GPU share)
python3 tools/soak_parity.py harvest25x38 4096 2000 250 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_harvest25x38_chains.log | tail -2
python3 tools/soak_parity.py cleanup48x36 2048 2000 250 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_cleanup48x36_chains.log | tail -2
python3 tools/soak_parity.py cleanup48x36 2048 1000 250 actions 2>&1 | grep -v amdgpu.ids | tee $D/soak_cleanup48x36_actions.log | tail -2
SSD_LIB_PATH=sequential_social_dilemma_games_amd/libssd_hip_testhooks.so SSD_AQL_ALTERNATE=1 python3 tools/soak_parity.py cleanup48x36 2048 1000 250 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_cleanup48x36_chains_alternate.log | tail -2
python3 tools/soak_parity.py harvest25x38 4096 1000 250 fused 2>&1 | grep -v amdgpu.ids | tee $D/soak_harvest25x38_fused.log | tail -2
# ... an observation ring beyond the memory-side cache (21 slots x 4096 envs = 290 MB: non-temporal write-back stores, a release once per
# round of the ring): EVERY step's observations and rewards compared in their slots after each 20-step call, 50 rounds of the ring
SOAK_RING=21 SOAK_CHECK_ALL=1 python3 tools/soak_parity.py harvest 4096 1000 20 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_harvest_ring21_every_step.log | tail -2
SOAK_RING=21 SOAK_CHECK_ALL=1 python3 tools/soak_parity.py cleanup 4096 600 20 actions 2>&1 | grep -v amdgpu.ids | tee $D/soak_cleanup_ring21_every_step_actions.log | tail -2
SSD_LIB_PATH=sequential_social_dilemma_games_amd/libssd_hip_testhooks.so SSD_AQL_ALTERNATE=1 SOAK_RING=21 SOAK_CHECK_ALL=1 python3 tools/soak_parity.py harvest 4096 600 20 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_harvest_ring21_every_step_alternate.log | tail -2
