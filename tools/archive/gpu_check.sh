#!/bin/bash
# tools/gpu_check.sh -- GPU box: the -m gpu suite, then 3 + 1 timing runs of bench.py (Harvest x3, Cleanup x1).
set -u
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-200
grep -q " failed" gpurun_out/pytest_gpu.log && exit 1
one() { python bench.py --steps 5000 --warmup 300 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s  %.2f us  %.3f G/s  frac %.3f' % (d['config']['workload'][:16], d['roofline']['avg_launch_us'], d['value']/1e9, d['roofline']['frac']))"; }
one; one; one; one --game cleanup
