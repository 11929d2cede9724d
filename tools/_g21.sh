cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
D=gpurun_out/r02h
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $D/pytest_gpu.log 2>&1; tail -3 $D/pytest_gpu.log
SSD_AQL_ALTERNATE=1 SSD_AQL_ALWAYS_FORK=1 timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "rollout or fullsize or vector or bench or enlarged" > $D/pytest_gpu_alt.log 2>&1; tail -3 $D/pytest_gpu_alt.log
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $D/bench_driver_$i.json 2> $D/bench_driver.err; done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver.json 2> $D/bench_driver.err; tail -2 $D/bench_driver.err
