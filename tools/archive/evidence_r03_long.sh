#!/bin/bash
# tools/evidence_r03_long.sh -- GPU box: the round's LAST build once more, longer: fuzz 1500 configurations (seed 1) and soaks of
# 4096 envs x 20 000 steps per game as rollout chains (82 M env-steps each), the enlarged Cleanup map 2048 envs x 5000 steps
# (10 agents: the two-pass beams), a 290-MB observation ring with every slot compared over 2000 steps.  Logs under profiles/r03_evidence/.
cd $GRAFT_REPO_ROOT
D=gpurun_out/r03_evidence_long
mkdir -p $D
python3 tools/fuzz_parity.py 1500 1 2>&1 | grep -v amdgpu.ids | tee $D/fuzz_1500_seed1.log | tail -2
for g in harvest cleanup; do
  python3 tools/soak_parity.py $g 4096 20000 1000 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_${g}_chains_20000.log | tail -1
done
python3 tools/soak_parity.py cleanup48x36 2048 5000 250 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_cleanup48x36_chains_5000.log | tail -1
python3 tools/soak_parity.py cleanup48x36 2048 2000 250 fused 2>&1 | grep -v amdgpu.ids | tee $D/soak_cleanup48x36_fused_2000.log | tail -1
python3 tools/soak_parity.py cleanup48x36 2048 2000 250 step 2>&1 | grep -v amdgpu.ids | tee $D/soak_cleanup48x36_step_2000.log | tail -1
SOAK_RING=21 SOAK_CHECK_ALL=1 python3 tools/soak_parity.py harvest 4096 2000 20 chains 2>&1 | grep -v amdgpu.ids | tee $D/soak_harvest_ring21_every_step_2000.log | tail -1
