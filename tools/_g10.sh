cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
D=gpurun_out/r02d
timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke.log 2>&1; echo "smoke rc=$?"; grep -i "assert\|error" $D/smoke.log | tail -2
SSD_AQL_ALTERNATE=1 SSD_AQL_ALWAYS_FORK=1 timeout -k 10 150 python3 tools/aql_smoke.py > $D/smoke_alt.log 2>&1; echo "smoke alt+fork rc=$?"; grep -i "assert\|error" $D/smoke_alt.log | tail -2
O=$D/ab6.txt
: > $O
run() { echo "== CH=${CH:-2} $*" >> $O; env "$@" timeout -k 10 100 python3 tools/rollout_modes.py 4096 ${CH:-2} 2>&1 | grep "n= 1000\|n= 3000\|n=   20\|n=  100" | tail -4 >> $O; }
run SSD_AQL=0
run SSD_AQL=1
run SSD_AQL=1 SSD_AQL_COHERENT=0
CH=3 run SSD_AQL=1
CH=1 run SSD_AQL=1
cat $O
timeout -k 10 100 python3 tools/short_call_breakdown.py > $D/short.txt 2>&1; cat $D/short.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $D/bench_driver.json 2> $D/bench_driver.err
timeout -k 10 300 python3 bench.py --no-configs --no-cpu-baseline > $D/bench_default.json 2> $D/bench_default.err
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $D/pytest_gpu.log 2>&1; tail -3 $D/pytest_gpu.log
