cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_hip_fullsize.py -q -x -m gpu -k "exact or enlarged" 2>&1 | tail -4
SSD_AQL_ALTERNATE=1 SSD_AQL_ALWAYS_FORK=1 timeout -k 10 600 python3 -m pytest tests/test_hip_fullsize.py -q -x -m gpu -k "exact or enlarged" 2>&1 | tail -4
