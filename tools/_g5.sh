cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
O=gpurun_out/r02b/ab2.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000" | tail -2 >> $O; }
run SSD_AQL=1
run SSD_AQL=1 SSD_AQL_SKIP_QUEUES=1
run SSD_AQL=1 SSD_AQL_SKIP_QUEUES=3
run SSD_AQL=1 SSD_AQL_PRIORITY=2
CH=4 run SSD_AQL=1 SSD_AQL_SKIP_QUEUES=1
CH=4 run SSD_AQL=1 SSD_AQL_SKIP_QUEUES=3
CH=4 run SSD_AQL=1 SSD_AQL_PRIORITY=2
CH=3 run SSD_AQL=1 SSD_AQL_REL=0
CH=4 run SSD_AQL=1 SSD_AQL_REL=0
CH=3 run SSD_AQL=1 SSD_AQL_SKIP_QUEUES=1 SSD_AQL_REL=0
cat $O
