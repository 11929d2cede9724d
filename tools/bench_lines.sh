#!/bin/bash
# tools/bench_lines.sh ROUND -- run ON THE GPU BOX: the bench lines kept under profiles/ for a round (default command, the driver's command x 3,
# the env-count sweep, the output-ring sweep).
ROUND=${1:?round tag, e.g. r04}
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/${ROUND}_final; mkdir -p $D
python3 tools/_label.py "bench_lines $ROUND" > $D/label.txt
python3 bench.py > $D/bench_default.json 2> $D/bench_default.err; echo "default rc=$?"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_$i.json 2> $D/bench_driver_$i.err; echo "driver $i rc=$?"; done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 5 --gather-leg --no-configs --no-cpu-baseline > $D/bench_tdr_1rank_gather.json 2> $D/bench_tdr.err; echo "tdr rc=$?"
: > $D/sweep_envs.txt
for E in 256 1024 2048 4096 8192 16384 32768 65536; do
  python3 bench.py --envs $E --steps 600 --warmup 100 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%6d envs: %7.2f us per step  %7.1f M agent-env-steps/s  frac %.3f  %s' % ($E, d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['frac'], d['config']['dispatch'][:48]))" | tee -a $D/sweep_envs.txt
done
: > $D/sweep_ring.txt
for R in 1 4 16 32 64; do
  python3 bench.py --ring $R --steps 1024 --warmup 128 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring %3d (%5d MB of observations): %6.2f us per step  frac %.3f' % ($R, $R*13824000//1000000, d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $D/sweep_ring.txt
done
