#!/usr/bin/env python3
"""Average each PMC counter per dispatch, per kernel, from a rocprofv3 counter_collection.csv."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
        acc[k][0] += float(row.get("Counter_Value", 0))
        acc[k][1] += 1
for (kern, ctr), (tot, n) in sorted(acc.items()):
    if "ssd" in kern:
        print("%-60s %-24s avg/dispatch %16.1f  (n=%d)" % (kern, ctr, tot / n, n))
