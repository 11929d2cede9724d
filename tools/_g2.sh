set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python3 tools/short_call_breakdown.py > gpurun_out/r02a/short_default.txt 2>&1
HSA_ENABLE_INTERRUPT=0 python3 tools/short_call_breakdown.py > gpurun_out/r02a/short_nointr.txt 2>&1
HSA_ENABLE_INTERRUPT=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/r02a/bench_driver_nointr.json 2> gpurun_out/r02a/bench_driver_nointr.err
