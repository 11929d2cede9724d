#!/bin/bash
# tools/matrix_c48.sh -- GPU box: Cleanup 48x36, 10 agents, 2048 envs (configs[4]'s per-GPU share) under the documented knobs:
# chains x split rendering x envs per workgroup.  us per step of a 600-step rollout, fresh process each.
set -u
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r03e}; mkdir -p $O
python tools/_label.py matrix_c48 > $O/matrix_c48.txt
for CH in 1 2; do for SP in 1 0; do for EPB in 0 2 4 8; do
  R=$(SSD_ROLLOUT_CHAINS=$CH SSD_AQL_SPLIT=$SP SSD_ENVS_PER_BLOCK=$EPB timeout -k 5 120 python bench.py --game cleanup48x36 --envs 2048 --steps 600 --warmup 100 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us  frac %.3f  %s' % (d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['dispatch'][:60]))")
  echo "chains $CH split $SP epb $EPB: $R" | tee -a $O/matrix_c48.txt
done; done; done
