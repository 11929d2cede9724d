#!/bin/bash
# tools/ab_r04.sh LIB_A LIB_B REPS TAG -- GPU box: library A against library B ("-" = the product library) in alternating fresh
# processes on the four single-GPU workloads (bench.py --no-extras, 600-step calls) and the driver's 20-step call of the
# headline; medians per workload into gpurun_out/ab/<TAG>.txt.
LIBA=$1; LIBB=$2; REPS=$3; TAG=$4
mkdir -p gpurun_out/ab
OUT=gpurun_out/ab/$TAG.txt
python3 tools/_label.py "ab_r04 A=$LIBA B=$LIBB reps=$REPS" > $OUT
for W in "--game harvest --envs 4096 --steps 600 --warmup 100" "--game cleanup48x36 --envs 2048 --steps 600 --warmup 100" \
         "--game cleanup --envs 4096 --steps 600 --warmup 100" "--game harvest25x38 --envs 4096 --steps 600 --warmup 100" \
         "--game harvest --envs 4096 --steps 20 --warmup 5"; do
  echo "== $W" >> $OUT
  bash tools/ab_lib.sh "$LIBA" "$LIBB" $REPS $W | tail -2 >> $OUT
done
cat $OUT
