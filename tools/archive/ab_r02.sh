#!/bin/bash
# GPU box: this tree against the round-2 tree (a git worktree under ab/r02, built in the container) at several env counts
cd "$GRAFT_REPO_ROOT"
for E in 8192 16384 32768; do
  for i in 1 2; do
    for T in . ab/r02; do
      for Q in "" "SSD_AQL_QUEUES=3"; do
        V=$(cd $T && env $Q python3 bench.py --envs $E --steps 600 --warmup 100 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), d['config']['dispatch'][:70].replace(' ','_'), d['config'].get('launches_per_step'))")
        echo "$E envs tree=$T $Q: $V"
      done
    done
  done
done
