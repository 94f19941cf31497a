#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402  (only for the fingerprint of the kernel sources)
print("kernel_source_sha: %s" % bench.kernel_source_sha())
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for row in csv.DictReader(open(f)):
        name = row["Name"].split("(")[0][-60:]
        print("%-60s calls %4s  avg %10.1f us  total %6.2f %%" %
              (name, row["Calls"], float(row["AverageNs"]) / 1e3, float(row["Percentage"])))
for f in glob.glob(os.path.join(root, "*.json")):
    print("== %s ==" % os.path.basename(f))
    print(open(f).read().strip()[:2000])
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== PMC %s (mean per dispatch) ==" % os.path.basename(d))
        for k, v in agg.items():
            if "sample_counts" not in k:
                continue
            for c, xs in sorted(v.items()):
                print("  %-40s %-26s %.6g  (n=%d)" % (k, c, sum(xs) / len(xs), len(xs)))
