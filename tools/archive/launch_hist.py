#!/usr/bin/env python3
"""Distribution of per-launch kernel durations from a rocprofv3 --kernel-trace csv (GPU box):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 3000 --no-cpu-baseline
    python3 tools/launch_hist.py $(find gpurun_out/kt -name '*kernel_trace.csv')"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"][:60] + " queue " + r.get("Queue_Id", "?"), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for name, v in by.items():
    if len(v) < 100:
        continue
    v.sort()
    st = np.array([a for a, _ in v], dtype=np.int64)
    en = np.array([b for _, b in v], dtype=np.int64)
    d = (en - st) / 1e3
    gap = (st[1:] - en[:-1]) / 1e3
    pitch = (st[1:] - st[:-1]) / 1e3
    q = [0, 10, 50, 90, 99, 100]
    print(name, "n=%d" % len(v))
    print("  duration us  pct %s: %s" % (q, np.percentile(d, q).round(2).tolist()))
    print("  gap us       pct %s: %s" % (q, np.percentile(gap, q).round(2).tolist()))
    print("  start pitch  pct %s: %s  mean %.2f" % (q, np.percentile(pitch, q).round(2).tolist(), pitch.mean()))

# overlap between queues: fraction of time with 0 / 1 / 2+ step kernels in flight
ev = []
for name, v in by.items():
    if "ssd_env_kernel" in name and len(v) >= 100:
        for a, b in v[len(v) // 4: -len(v) // 4]:
            ev.append((a, 1)); ev.append((b, -1))
ev.sort()
if ev:
    busy = {}
    cur, last = 0, ev[0][0]
    for t, d in ev:
        busy[cur] = busy.get(cur, 0) + (t - last)
        cur += d; last = t
    tot = sum(busy.values())
    print("step kernels in flight (middle half of the run): " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(busy.items())))
