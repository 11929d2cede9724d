cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
D=gpurun_out/r02g
O=$D/ab12.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 3000\|n=   20" | tail -2 >> $O; }
run SSD_SPLIT_PRIO=0
run SSD_SPLIT_PRIO=1
run SSD_SPLIT_PRIO=2
run SSD_SPLIT_PRIO=3
CH=3 run SSD_SPLIT_PRIO=2 SSD_SPLIT_EPB=2
CH=3 run SSD_SPLIT_PRIO=2
run SSD_SPLIT_PRIO=2 SSD_SPLIT_EPB=2
cat $O
