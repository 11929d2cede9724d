#!/bin/bash
# tools/phase_r03.sh -- GPU box: the per-phase profiles of the round's final build (diagnostic library, 2 chains), kept under profiles/
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r03_phase; mkdir -p $D
export SSD_PROFILE_CHAINS=2 SSD_LIB_PATH=sequential_social_dilemma_games_amd/libssd_hip_stamps.so
timeout -k 5 200 python3 tools/phase_profile.py harvest 4096 > $D/phase_harvest.txt 2>&1; echo "harvest rc=$?"
timeout -k 5 200 python3 tools/phase_profile.py cleanup 4096 > $D/phase_cleanup.txt 2>&1; echo "cleanup rc=$?"
timeout -k 5 200 python3 tools/phase_profile.py cleanup48x36 2048 > $D/phase_cleanup48x36.txt 2>&1; echo "cleanup48x36 rc=$?"
