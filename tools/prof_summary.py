#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (kernel stats + PMC csv) into text."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, root))
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 12:
            print("  {Name:.90s} calls={Calls} avg_ns={AverageNs} total_ns={TotalDurationNs} pct={Percentage}".format(**row))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")[:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
    print("== pmc:", os.path.basename(d))
    for k, cs in agg.items():
        if "fir" in k or "demod" in k or "mfma" in k or "fused" in k:
            print("  ", k)
            for c, v in cs.items():
                n = cnt[(k, c)]
                print(f"      {c:28s} total={v:.4g}  per_dispatch={v / n:.6g}  (n={n})")
