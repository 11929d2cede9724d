cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
D=gpurun_out/r02e
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $D/pytest_gpu.log 2>&1; tail -15 $D/pytest_gpu.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver.json 2> $D/bench_driver.err; tail -2 $D/bench_driver.err
