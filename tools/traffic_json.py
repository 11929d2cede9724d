#!/usr/bin/env python3
"""profiles/<round>_traffic.json (python tools/traffic_json.py r04) from the rocprofv3 summaries tools/profile_workloads.sh <round> leaves in gpurun_out/<round>_prof/<tag>/:
HBM bytes per step = (2 x FETCH_SIZE + WRITE_SIZE) KiB of the step kernel's dispatches (separate --pmc passes; FETCH_SIZE doubled per
the gfx950 correction of MI355X_MICROARCH.md) x the launches of one step, next to the algorithmic bytes of SURVEY.md 8d; plus
the step kernel's average duration under --kernel-trace.  Also copies the per-workload summaries into profiles/."""
import csv
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from sequential_social_dilemma_games_amd import config as cfgmod  # noqa: E402

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = os.path.join(REPO, "gpurun_out", "%s_prof" % ROUND)
DST = os.path.join(REPO, "profiles")
WORK = {   # tag -> (H, W, N, E, chains, f32)
    "harvest_16x38_n5_e4096": (16, 38, 5, 4096, 2, False), "cleanup_25x18_n5_e4096": (25, 18, 5, 4096, 2, False),
    "harvest_25x38_n5_e4096": (25, 38, 5, 4096, 2, False), "cleanup_48x36_n10_e2048": (48, 36, 10, 2048, 2, False),
    "harvest_16x38_n5_e4096_f32": (16, 38, 5, 4096, 2, True),
}


def step_kernel(name):
    """ssd_env_kernel<GAME, MODE = 0 (step), ...>: the launches that make up a step."""
    m = re.search(r"ssd_env_kernel<(\d), (\d),", name)
    return bool(m) and m.group(2) == "0"


def main():
    out = {}
    for tag, (H, W, N, E, chains, f32) in WORK.items():
        d = os.path.join(SRC, tag)
        if not os.path.isdir(d):
            continue
        ent = {"launches_per_step": chains}
        ks = os.path.join(d, "kernel_stats.csv")
        if os.path.exists(ks):
            shutil.copy(ks, os.path.join(DST, "%s_%s_kernel_stats.csv" % (ROUND, tag)))
            rows = [r for r in csv.DictReader(open(ks)) if step_kernel(r["Name"])]
            if rows:
                r = max(rows, key=lambda r: int(r["Calls"]))
                ent["rocprof_avg_kernel_us"] = float(r["AverageNs"]) / 1e3
                ent["rocprof_kernel"] = r["Name"].split("(")[0]
                ent["rocprof_calls"] = int(r["Calls"])
                ent["rocprof_source"] = ("profiles/%s_%s_kernel_stats.csv: per launch of %d envs under rocprofv3 --kernel-trace (tracing adds "
                                         "per-dispatch overhead; not the unprofiled step time)" % (ROUND, tag, E // chains))
        ps = os.path.join(d, "pmc_summary.txt")
        if os.path.exists(ps) and os.path.getsize(ps):
            shutil.copy(ps, os.path.join(DST, "%s_%s_pmc_summary.txt" % (ROUND, tag)))
            vals = {}
            for line in open(ps):
                m = re.match(r"(.*?)\s+(\S+)\s+avg/dispatch\s+([\d.]+)\s+\(n=(\d+)\)", line)
                if m and step_kernel(m.group(1)) and m.group(2) in ("FETCH_SIZE", "WRITE_SIZE"):
                    k = m.group(2)
                    if k not in vals or int(m.group(4)) > vals[k][1]:
                        vals[k] = (float(m.group(3)), int(m.group(4)))
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
                alg = cfgmod.algorithmic_bytes_per_env_step(H, W, N, 15) + (N * 15 * 15 * 3 * 3 if f32 else 0)
                ent.update(fetch_size_kib_per_launch=f, write_size_kib_per_launch=w,
                           hbm_bytes_per_launch=int(round((2 * f + w) * 1024 * chains)), algorithmic_bytes_per_step=alg * E)
                ent["ratio"] = round(ent["hbm_bytes_per_launch"] / ent["algorithmic_bytes_per_step"], 3)
                ent["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_workloads.sh %s; host-side waits, chosen by the "
                               "library itself under a tool); FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM).  bench.py steps the "
                               "envs as %d concurrent launch(es) per step; the counters are per launch, hbm_bytes_per_launch here is the sum over the "
                               "launches of one step." % (ROUND, chains))
        bt = os.path.join(d, "bench_under_trace.json")
        if os.path.exists(bt) and os.path.getsize(bt):
            try:
                ent["us_per_step_under_kernel_trace"] = json.load(open(bt))["ms_per_step"] * 1e3
            except Exception:
                pass
        out[tag] = ent
    json.dump(out, open(os.path.join(DST, "%s_traffic.json" % ROUND), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
