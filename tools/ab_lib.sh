#!/bin/bash
# tools/ab_lib.sh LIB_A LIB_B REPS bench-args... -- on the GPU box: bench.py --no-extras [args] with library A and library B
# (SSD_LIB_PATH; "-" = the product library), alternating fresh processes on the same box; us per step of each run and the medians.
LIBA=$1; LIBB=$2; REPS=$3; shift; shift; shift
run() {
  if [ "$1" = "-" ]; then env -u SSD_LIB_PATH python3 bench.py --no-extras "${@:3}" 2>/dev/null; else SSD_LIB_PATH=$1 python3 bench.py --no-extras "${@:3}" 2>/dev/null; fi |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(d['ms_per_step']*1e3,3))"
}
for i in $(seq 1 $REPS); do
  run "$LIBA" A "$@"
  run "$LIBB" B "$@"
done | tee /tmp/ab_lib.txt
python3 - <<PY
import collections
d=collections.defaultdict(list)
for l in open('/tmp/ab_lib.txt'):
    k,v=l.split(); d[k].append(float(v))
for k,v in sorted(d.items()):
    v.sort(); print(k, 'median', v[len(v)//2], 'min', v[0], 'max', v[-1])
PY
