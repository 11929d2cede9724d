cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
D=gpurun_out/r02h
for i in 1 2 3 4; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $D/bench_driver_$i.json 2> $D/bench_driver.err; done
for i in 5 6; do SSD_AQL_FORK_SPIN_US=0 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $D/bench_driver_$i.json 2> $D/bench_driver.err; done
python3 tools/bench_region_probe.py 2>&1 | grep -v amdgpu
python3 tools/bench_region_probe.py 2>&1 | grep -v amdgpu
