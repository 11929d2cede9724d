#!/bin/bash
# tools/ab_short_calls.sh VAR A B [reps] -- on the GPU box: the driver's bench command with VAR=A and VAR=B alternating (fresh
# processes, same box), the us per step of each run and the medians.
VAR=$1; A=$2; B=$3; REPS=${4:-5}
for i in $(seq 1 $REPS); do
  for V in $A $B; do
    env $VAR=$V python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$V', round(d['ms_per_step']*1e3,3))"
  done
done | tee /tmp/ab.txt
python3 - <<PY
import collections
d=collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    k,v=l.split(); d[k].append(float(v))
for k,v in d.items():
    v.sort(); print(k, 'median', v[len(v)//2], 'min', v[0], 'max', v[-1])
PY
