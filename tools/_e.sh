cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -q -x -m gpu -k "rollout or fullsize or golden or envs" 2>&1 | tail -2
for i in 1 2; do python3 tools/rollout_modes.py 4096 2 2>&1 | grep "n= 3000\|n= 1000"; done
