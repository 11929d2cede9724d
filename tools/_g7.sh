cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
D=gpurun_out/r02c
O=$D/ab4.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000" | tail -2 >> $O; }
run SSD_AQL=0
run SSD_AQL=1 SSD_AQL_COHERENT=0
run SSD_AQL=1 SSD_AQL_COHERENT=0 SSD_AQL_SIGNAL=1
run SSD_AQL=1 SSD_AQL_COHERENT=0 SSD_AQL_SIGNAL=1 SSD_AQL_QUEUE_MULTI=1
run SSD_AQL=1 SSD_AQL_COHERENT=1 SSD_AQL_SIGNAL=1
run SSD_AQL=1 SSD_AQL_COHERENT=0 SSD_AQL_REL=0 SSD_AQL_SIGNAL=1
cat $O
