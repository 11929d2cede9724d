#!/bin/bash
# tools/ab_env.sh VAR A B REPS bench-args... -- on the GPU box: bench.py --no-extras [args] with VAR=A and VAR=B alternating
# (fresh processes, same box); us per step of each run and the medians.
VAR=$1; A=$2; B=$3; REPS=$4; shift; shift; shift; shift
for i in $(seq 1 $REPS); do
  for V in $A $B; do
    env $VAR=$V python3 bench.py --no-extras "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$V', round(d['ms_per_step']*1e3,3))"
  done
done | tee /tmp/ab_env.txt
python3 - <<PY
import collections
d=collections.defaultdict(list)
for l in open('/tmp/ab_env.txt'):
    k,v=l.split(); d[k].append(float(v))
for k,v in sorted(d.items()):
    v.sort(); print(k, 'median', v[len(v)//2], 'min', v[0], 'max', v[-1])
PY
