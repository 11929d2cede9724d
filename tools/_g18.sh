cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
D=gpurun_out/r02g
O=$D/ab11.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 3000\|n=   20" | tail -2 >> $O; }
for ch in 2 3; do for epb in 1 2 4 8; do CH=$ch run SSD_ENVS_PER_BLOCK=$epb; done; done
CH=2 run SSD_ENVS_PER_BLOCK=4 SSD_AQL_SPLIT=0
CH=3 run SSD_ENVS_PER_BLOCK=4 SSD_AQL_SPLIT=0
cat $O
