cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
D=gpurun_out/r02e
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $D/pytest_gpu.log 2>&1; tail -25 $D/pytest_gpu.log
