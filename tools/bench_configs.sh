for ARGS in "--game harvest --envs 4096" "--game cleanup --envs 4096" "--game harvest25x38 --envs 4096" "--game cleanup48x36 --envs 2048" "--game cleanup48x36 --envs 4096"; do
  timeout -k 10 180 python bench.py $ARGS --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-70s %7.2f us  %8.1f M/s  frac %.3f' % (d['config']['workload'][:70], d['roofline']['avg_launch_us'], d['value']/1e6, d['roofline']['frac']))"
done
# the same with pipelined launches where the batch is small enough for them (SSD_ROLLOUT_PIPELINED, two output slots)
for ARGS in "--game harvest --envs 2048" "--game cleanup --envs 2048" "--game cleanup48x36 --envs 2048" "--game harvest --envs 1024"; do
  for PL in "" "--pipelined"; do
    timeout -k 10 180 python bench.py $ARGS --ring 2 $PL --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-58s %-11s %7.2f us  %8.1f M/s  frac %.3f' % (d['config']['workload'][:58], '$PL', d['roofline']['avg_launch_us'], d['value']/1e6, d['roofline']['frac']))"
  done
done
