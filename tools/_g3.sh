set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout -k 10 120 python3 tools/rollout_modes.py 4096 2 > gpurun_out/r02b/modes_aql_c2.txt 2>&1
timeout -k 10 120 python3 tools/rollout_modes.py 4096 3 > gpurun_out/r02b/modes_aql_c3.txt 2>&1
timeout -k 10 120 python3 tools/rollout_modes.py 4096 4 > gpurun_out/r02b/modes_aql_c4.txt 2>&1
timeout -k 10 120 python3 tools/rollout_modes.py 4096 1 > gpurun_out/r02b/modes_aql_c1.txt 2>&1
timeout -k 10 120 python3 tools/short_call_breakdown.py > gpurun_out/r02b/short_aql.txt 2>&1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/r02b/bench_driver.json 2> gpurun_out/r02b/bench_driver.err
timeout -k 10 300 python3 bench.py --no-configs --no-cpu-baseline > gpurun_out/r02b/bench_default.json 2> gpurun_out/r02b/bench_default.err
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02b/pytest_gpu.log 2>&1
tail -3 gpurun_out/r02b/pytest_gpu.log
