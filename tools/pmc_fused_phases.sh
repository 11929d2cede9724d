#!/bin/bash
# tools/pmc_fused_phases.sh [harvest|cleanup] [E] -- GPU box, diagnostic library (make stamps): dynamic instructions per wave and STEP
# of the fused rollout kernel with each phase skipped in turn (differences against the first line = that phase's instructions).
# Output: gpurun_out/pmc_roles/fused_phases_<game>.txt
set -u
GAME=${1:-harvest}; E=${2:-4096}; STEPS=200
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_LIB_PATH=$GRAFT_REPO_ROOT/sequential_social_dilemma_games_amd/libssd_hip_stamps.so PMC_FUSED=1
O=gpurun_out/pmc_roles; mkdir -p $O
python3 tools/_label.py "pmc_fused_phases $GAME $E" > $O/fused_phases_$GAME.txt
for CFG in "0 full" "1 no-move" "2 no-consume" "4 no-beams" "8 no-respawn" "32 no-overlay" "16 obs-to-64-envs" "47 floor"; do
  set -- $CFG
  rm -rf $O/run
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR --output-format csv -d $O/run -- python3 tools/pmc_roles_run.py $1 $GAME $E $STEPS > $O/phases.log 2>&1
  F=$(find $O/run -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$2" $STEPS >> $O/fused_phases_$GAME.txt <<'PY'
import csv, sys, collections
by = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "ssd_env_kernel" in r["Kernel_Name"] and ", 3, false" in r["Kernel_Name"][:40]:
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
steps = int(sys.argv[3])
ds = sorted((d for d in by.values() if "SQ_WAVES" in d), key=lambda d: -d["SQ_INSTS_VALU"])[:3]      # the long launches
n = len(ds)
print("%-18s per wave and step: " % sys.argv[2] + "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", ""), sum(d[k] / d["SQ_WAVES"] for d in ds) / n / steps)
                                                  for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_WR")) + "  (%d launches of %d steps)" % (n, steps))
PY
  rm -rf $O/run
done
cat $O/fused_phases_$GAME.txt
