#!/bin/bash
# tools/pmc_roles_phases.sh [harvest|cleanup|cleanup48x36] [E] -- GPU box, diagnostic library (make stamps): dynamic instructions per
# ENV wave of a split rollout's launches with each phase skipped in turn (differences against the first line = that phase's
# instructions; the env-only launches are the first launch of every call).  Output: gpurun_out/pmc_roles/phases_<game>.txt
set -u
GAME=${1:-harvest}; E=${2:-4096}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SSD_LIB_PATH=$GRAFT_REPO_ROOT/sequential_social_dilemma_games_amd/libssd_hip_stamps.so
O=gpurun_out/pmc_roles; mkdir -p $O
python3 tools/_label.py "pmc_roles_phases $GAME $E" > $O/phases_$GAME.txt
for CFG in "0 full" "1 no-move" "2 no-consume" "4 no-beams" "8 no-respawn" "15 floor"; do
  set -- $CFG
  rm -rf $O/run
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $O/run -- python3 tools/pmc_roles_run.py $1 $GAME $E > $O/phases.log 2>&1
  F=$(find $O/run -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$2" >> $O/phases_$GAME.txt <<'PY'
import csv, sys, collections
by = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "ssd_env_kernel" in r["Kernel_Name"] and ", 0, false" in r["Kernel_Name"][:40]:
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
ds = [d for d in by.values() if "SQ_WAVES" in d]
full = max(d["SQ_WAVES"] for d in ds)
one = [d for d in ds if d["SQ_WAVES"] < 0.75 * full]           # launches of ONE role
one.sort(key=lambda d: d["SQ_INSTS_LDS"] / d["SQ_WAVES"])      # env waves: fewer LDS instructions than renderer waves
env = one[: len(one) // 2]
n = len(env)
print("%-12s env wave: " % sys.argv[2] + "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", ""), sum(d[k] / d["SQ_WAVES"] for d in env) / n)
                                                  for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM")) + "  (%d launches)" % n)
PY
  rm -rf $O/run
done
cat $O/phases_$GAME.txt
