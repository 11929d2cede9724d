#!/bin/bash
# tools/final_r02.sh -- run ON THE GPU BOX: the round's bench lines (default command, the driver's command three times, the env-count
# sweep) into gpurun_out/r02_final/.
cd $GRAFT_REPO_ROOT
D=gpurun_out/r02_final
mkdir -p $D
python3 bench.py > $D/bench_default.json 2> $D/bench_default.err; echo "default rc=$?"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_$i.json 2> $D/bench_driver_$i.err; echo "driver $i rc=$?"; done
: > $D/sweep_envs.txt
for E in 256 1024 2048 4096 8192 16384 32768 65536; do
  python3 bench.py --envs $E --steps 2000 --warmup 200 --no-extras 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('E=%6d  %7.2f us/step  %8.1f M agent-steps/s  frac %.3f  (%s)' % ($E, r['ms_per_step']*1e3, r['value']/1e6, r['roofline']['frac'], r['config']['enqueue']))" >> $D/sweep_envs.txt
done
cat $D/sweep_envs.txt
