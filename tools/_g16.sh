cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
D=gpurun_out/r02g
timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_split.log 2>&1; echo "smoke split rc=$?"; grep -i "assert\|error" $D/smoke_split.log | tail -2
SSD_AQL_ALTERNATE=1 SSD_AQL_ALWAYS_FORK=1 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_split_alt.log 2>&1; echo "smoke split alt rc=$?"; grep -i "assert\|error" $D/smoke_split_alt.log | tail -2
SSD_AQL_SPLIT=0 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_nosplit.log 2>&1; echo "smoke nosplit rc=$?"
O=$D/ab9.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000\|n=   20\|n=  100" | tail -4 >> $O; }
run SSD_AQL_SPLIT=0
run SSD_AQL_SPLIT=1
CH=1 run SSD_AQL_SPLIT=1
CH=3 run SSD_AQL_SPLIT=1
run SSD_AQL_SPLIT=1 SSD_ENVS_PER_BLOCK=4
cat $O
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $D/pytest_gpu.log 2>&1; tail -4 $D/pytest_gpu.log
