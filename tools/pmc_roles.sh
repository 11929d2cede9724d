#!/bin/bash
# tools/pmc_roles.sh TAG [bench args] -- GPU box: dynamic instructions per wave of the two ROLES of a split rollout's launches.
# One rocprofv3 --pmc pass of a short bench.py run; the dispatches of the step kernel are then told apart by their wave counts and
# instruction totals: launches of env waves alone (the first step of a call), of renderer waves alone (the launch that ends a
# call) and of both (every step between).  Output: gpurun_out/pmc_roles/<TAG>.txt
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_roles; mkdir -p $O
python3 tools/_label.py "pmc_roles $TAG $*" > $O/$TAG.txt
# (PMC_LIST="SQ_WAVES SQ_INSTS_VALU ..." in the environment: other counters, at most eight SQ ones per pass; SQ_WAVES and SQ_INSTS_VALU tell the roles apart)
for PMC in "${PMC_LIST:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD}"; do
  rm -rf $O/run
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/run -- python3 bench.py --steps 40 --warmup 8 --no-extras "$@" > $O/$TAG.log 2>&1
  echo "rc=$?"
  F=$(find $O/run -name "*counter_collection.csv" | head -1)
  python3 - "$F" >> $O/$TAG.txt <<'PY'
import csv, sys, collections
by = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "ssd_env_kernel" in r["Kernel_Name"] and ", 0, false" in r["Kernel_Name"][:40]:
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        by[r["Dispatch_Id"]]["name"] = r["Kernel_Name"][:64]
groups = collections.defaultdict(list)
for d in by.values():
    if "SQ_WAVES" in d:
        groups[(d["name"], int(round(d["SQ_WAVES"] / 64.0)) * 64)].append(d)
for (name, waves), ds in sorted(groups.items()):
    # the 1-role launches come in two kinds: split them by their vector instructions per wave
    ds.sort(key=lambda d: d.get("SQ_INSTS_VALU", 0))
    parts = [ds]
    if len(ds) > 3 and ds[-1].get("SQ_INSTS_VALU", 0) > 1.25 * ds[0].get("SQ_INSTS_VALU", 1):
        mid = (ds[0]["SQ_INSTS_VALU"] + ds[-1]["SQ_INSTS_VALU"]) / 2
        parts = [[d for d in ds if d["SQ_INSTS_VALU"] < mid], [d for d in ds if d["SQ_INSTS_VALU"] >= mid]]
    for part in parts:
        n = len(part)
        w = sum(d["SQ_WAVES"] for d in part) / n
        print("%s  ~%d waves per launch, %d launches: per wave " % (name, waves, n) +
              "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", ""), sum(d.get(k, 0) for d in part) / n / w)
                        for k in sorted(set().union(*[set(d) for d in part]) - {"name", "SQ_WAVES"})))
PY
  rm -rf $O/run
done
cat $O/$TAG.txt
