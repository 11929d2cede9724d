#!/bin/bash
# tools/ab_bench.sh libA.so libB.so [bench args] -- alternate two builds of the library on the SAME box (GPU boxes differ by a few %).
A=$1; B=$2; shift 2
one() { SSD_LIB_PATH=$1 python bench.py --steps 4000 --warmup 300 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f (fused %.2f)' % (d['roofline']['avg_launch_us'], d.get('fused_rollout', {}).get('us_per_step', 0.0)), end=' ')"; }
for i in 1 2 3; do
  echo -n "A: "; one $PWD/$A "$@"; echo -n "  B: "; one $PWD/$B "$@"; echo
done
