cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
O=gpurun_out/r02b/ab.txt
: > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000\|n=   20" | tail -3 >> $O; }
run SSD_AQL=0
run SSD_AQL=1
run SSD_AQL=0
run SSD_AQL=1
run SSD_AQL=1 SSD_AQL_QUEUE_MULTI=1
run SSD_AQL=1 SSD_AQL_REL=0
run SSD_AQL=1 SSD_AQL_ACQ=0
run SSD_AQL=1 SSD_AQL_ACQ=0 SSD_AQL_REL=0
run SSD_AQL=1 SSD_AQL_ACQ=0 SSD_AQL_REL=0 SSD_AQL_BARRIER=0
run SSD_AQL=1 SSD_AQL_ACQ=2 SSD_AQL_REL=2
CH=1 run SSD_AQL=0
CH=1 run SSD_AQL=1
CH=4 run SSD_AQL=0
CH=4 run SSD_AQL=1
CH=4 run SSD_AQL=1 GPU_MAX_HW_QUEUES=1
CH=3 run SSD_AQL=1 GPU_MAX_HW_QUEUES=1
CH=2 run SSD_AQL=1 GPU_MAX_HW_QUEUES=1
cat $O
