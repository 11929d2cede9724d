#!/bin/bash
# GPU box: 2 against 3 chains (the pool lifted to three queues) at the named sizes
cd "$GRAFT_REPO_ROOT"
python3 tools/_label.py chains3
for CFG in "--game harvest --envs 4096" "--game cleanup48x36 --envs 2048" "--game cleanup --envs 4096" "--game harvest25x38 --envs 4096"; do
  for CH in 2 3; do
    for i in 1 2 3; do
      V=$(SSD_AQL_QUEUES=3 SSD_ROLLOUT_CHAINS=$CH python3 bench.py --no-extras $CFG --steps 1000 --warmup 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,3), d['config']['dispatch'][:40].replace(' ','_'))")
      echo "$CFG chains $CH: $V"
    done
  done
done
