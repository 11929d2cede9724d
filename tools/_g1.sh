set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
SSD_ROLLOUT_THREADS=0 python3 tools/rollout_modes.py > gpurun_out/r02a/modes_inline.txt 2>&1
SSD_ROLLOUT_THREADS=1 python3 tools/rollout_modes.py > gpurun_out/r02a/modes_threads.txt 2>&1
python3 tools/rollout_modes.py > gpurun_out/r02a/modes_auto.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver.json 2> gpurun_out/r02a/bench_driver.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/r02a/bench_driver2.json 2>> gpurun_out/r02a/bench_driver.err
python3 bench.py --no-configs --no-cpu-baseline > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
python3 -m pytest tests -x -q -m gpu > gpurun_out/r02a/pytest_gpu.log 2>&1
tail -3 gpurun_out/r02a/pytest_gpu.log
