cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
D=gpurun_out/r02g
O=$D/ab10.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000" | tail -2 >> $O; }
run SSD_AQL_SPLIT=0
run SSD_AQL_SPLIT=0 SSD_EXP_ONE_QUEUE=1
CH=4 run SSD_AQL_SPLIT=0 SSD_EXP_ONE_QUEUE=1
CH=8 run SSD_AQL_SPLIT=0 SSD_EXP_ONE_QUEUE=1
cat $O
