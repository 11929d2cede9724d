#!/bin/bash
# tools/ab_multi.sh REPS "bench args" LIB1 LIB2 ... -- GPU box: bench.py --no-extras <args> with each library in turn (SSD_LIB_PATH; "-" =
# the product), REPS rounds of alternating fresh processes; us per step per run and the medians.  First line: what ran (tools/_label.py).
REPS=$1; ARGS=$2; shift; shift
cd "$GRAFT_REPO_ROOT"
python3 tools/_label.py "ab_multi $REPS [$ARGS] $*"
OUT=$(mktemp)
for i in $(seq 1 $REPS); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then V=$(env -u SSD_LIB_PATH python3 bench.py --no-extras $ARGS 2>/dev/null); else V=$(SSD_LIB_PATH=$L python3 bench.py --no-extras $ARGS 2>/dev/null); fi
    echo "$V" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step']*1e3,3))" | tee -a $OUT
  done
done
python3 - <<PY
import collections
d=collections.defaultdict(list)
for l in open('$OUT'):
    k,v=l.split(); d[k].append(float(v))
for k,v in d.items():
    v.sort(); print('MEDIAN %-70s %.3f  (min %.3f max %.3f, n=%d)' % (k, v[len(v)//2], v[0], v[-1], len(v)))
PY
