cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
D=gpurun_out/r02c
timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_coh.log 2>&1; echo "smoke coh rc=$?"
SSD_AQL_ALTERNATE=1 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_alt.log 2>&1; echo "smoke alt rc=$?"
SSD_AQL_ALTERNATE=1 SSD_AQL_ACQ=0 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_alt_acq0.log 2>&1; echo "smoke alt acq0 rc=$?"
SSD_AQL_COHERENT=0 SSD_AQL_REL=0 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_plain_rel0.log 2>&1; echo "smoke plain rel0 (expected to FAIL if release matters) rc=$?"
SSD_AQL_ALWAYS_FORK=1 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_fork.log 2>&1; echo "smoke always-fork rc=$?"
O=$D/ab3.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000\|n=   20" | tail -3 >> $O; }
run SSD_AQL=0
run SSD_AQL=1
run SSD_AQL=1 SSD_AQL_COHERENT=0
run SSD_AQL=1 SSD_AQL_ACQ=0
CH=3 run SSD_AQL=1
CH=1 run SSD_AQL=1
cat $O
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $D/bench_driver.json 2> $D/bench_driver.err
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $D/pytest_gpu.log 2>&1; tail -3 $D/pytest_gpu.log
SSD_AQL_ALTERNATE=1 timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "rollout or fullsize or vector" > $D/pytest_gpu_alt.log 2>&1; tail -3 $D/pytest_gpu_alt.log
