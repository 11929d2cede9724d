#!/bin/bash
# tools/sweep_envs.sh -- bench.py over env counts (run on the GPU box): separates the fixed per-launch cost
# from the per-env cost of the fused step kernel.
for GAME in harvest cleanup; do
  for E in ${ENVS:-256 1024 2048 4096 8192 16384 32768 65536 262144}; do
    timeout -k 10 180 python bench.py --game $GAME --envs $E --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s E=%7d  %8.2f us/step  %8.1f M agent-steps/s  roofline %.3f' % ('$GAME', d['config']['envs_per_gpu'], d['roofline']['avg_launch_us'], d['value']/1e6, d['roofline']['frac']))"
  done
done
