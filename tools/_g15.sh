cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
D=gpurun_out/r02f
O=$D/ab8.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000\|n=   20" | tail -3 >> $O; }
run SSD_AQL=1
run SSD_EXP_NO_OBS=1
CH=1 run SSD_EXP_NO_OBS=1
CH=3 run SSD_EXP_NO_OBS=1
run SSD_AQL_SYNC=1
cat $O
SSD_AQL_SYNC=1 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_sync.log 2>&1; echo "smoke sync rc=$?"
