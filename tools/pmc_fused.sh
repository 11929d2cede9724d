#!/bin/bash
# tools/pmc_fused.sh -- GPU box: is the fused rollout kernel bound by the CU's ONE scalar unit?  Counter passes over tools/fused_short.py
# (4096 Harvest envs; the 1000-step launch is the one to read: per-dispatch sums over the device, 256 CUs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_fused; mkdir -p $O
python3 tools/_label.py pmc_fused > $O/summary.txt
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/p$i -- python3 tools/fused_short.py > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
  F=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 - "$F" >> $O/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# the longest dispatches of the rollout kernel (MODE 3) = the 1000-step launches
by = collections.defaultdict(dict)
for r in rows:
    if "ssd_env_kernel<0, 3" in r["Kernel_Name"]:
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
best = sorted(by.values(), key=lambda d: -d.get("SQ_WAVE_CYCLES", d.get("SQ_ACTIVE_INST_ANY", 0)))[:3]
for d in best[:1]:
    for k, v in sorted(d.items()):
        print("%-24s %16.0f   per step (/1000) %12.1f   per CU and step %10.1f" % (k, v, v / 1000, v / 1000 / 256))
PY
  rm -rf $O/p$i
done
cat $O/summary.txt
