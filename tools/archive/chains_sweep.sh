#!/bin/bash
# tools/chains_sweep.sh GAME ENVS STEPS -- on the GPU box: us per step of bench.py --no-extras for 1, 2 and 3 chains
for c in 1 2 3; do
  python3 bench.py --no-extras --game $1 --envs $2 --chains $c --steps $3 --warmup 100 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 envs, chains $c:', round(d['ms_per_step']*1e3,3), 'us', d['config']['dispatch'])"
done
