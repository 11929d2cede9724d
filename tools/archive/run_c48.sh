#!/bin/bash
# tools/run_c48.sh [outdir] -- GPU box: parity of the Cleanup 48x36 kernels, then its timing (3 fresh processes) and phase profile
set -u
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r03f}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "enlarged or replayed or last_ranks or fuzz_slice or dispatch_modes" > $O/pytest_c48.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_c48.log
for i in 1 2 3; do
  timeout -k 5 120 python bench.py --game cleanup48x36 --envs 2048 --steps 600 --warmup 100 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c48 x 2048: %.2f us  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $O/c48_times.txt
done
timeout -k 5 120 python bench.py --game cleanup --envs 4096 --steps 600 --warmup 100 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cleanup 25x18 x 4096: %.2f us  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $O/c48_times.txt
timeout -k 5 120 python bench.py --steps 600 --warmup 100 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('harvest x 4096: %.2f us  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $O/c48_times.txt
SSD_PROFILE_CHAINS=2 SSD_LIB_PATH=sequential_social_dilemma_games_amd/libssd_hip_stamps.so timeout -k 5 200 python tools/phase_profile.py cleanup48x36 2048 > $O/phase_c48.txt 2>&1; tail -32 $O/phase_c48.txt
