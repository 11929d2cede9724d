cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
D=gpurun_out/r02f
O=$D/ab7.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000\|n=   20" | tail -3 >> $O; }
run SSD_AQL=1
run SSD_LIB_PATH=$PWD/sequential_social_dilemma_games_amd/libssd_exp_noendwait.so
run SSD_AQL_ACQ=0
run SSD_LIB_PATH=$PWD/sequential_social_dilemma_games_amd/libssd_exp_noendwait.so SSD_AQL_ACQ=0
run SSD_ENVS_PER_BLOCK=16
run SSD_ENVS_PER_BLOCK=4
run SSD_ENVS_PER_BLOCK=2
run SSD_ENVS_PER_BLOCK=1
run SSD_AQL=1
cat $O
