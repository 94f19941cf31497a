#!/usr/bin/env python3
"""Per-kernel durations and the gaps between the kernels of a sample_counts call, from a rocprofv3 kernel trace of
tools/kbench_ab.py (burst mode: calls back to back).  usage: tools/gap_trace.py <dir with *kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = {"prep_kernel": "prep", "sample_counts_stream_kernel": "stream", "sample_counts_heavy_kernel": "K3h", "row_flags_kernel": "rowflags"}
seq = []
for r in rows:
    for k, v in names.items():
        if k in r["Kernel_Name"]:
            seq.append((v, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, (n, s, e) in enumerate(seq):
    dur[n].append((e - s) / 1e3)
    if i:
        pn, ps, pe = seq[i - 1]
        if (s - pe) < 50000:       # inside a burst
            gap[pn + "->" + n].append((s - pe) / 1e3)
med = lambda xs: sorted(xs)[len(xs) // 2]
for n, xs in dur.items():
    print("kernel %-9s n=%4d median %8.2f us" % (n, len(xs), med(xs)))
for n, xs in gap.items():
    print("gap    %-16s n=%4d median %6.2f us" % (n, len(xs), med(xs)))
calls = [seq[i + 2][2] - seq[i][1] for i in range(len(seq) - 2) if [x[0] for x in seq[i:i + 3]] == ["prep", "stream", "K3h"]]
if calls:
    print("call (prep start .. K3h end) median %.2f us" % (med(calls) / 1e3))
