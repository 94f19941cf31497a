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
# the K timed dispatches of the stream kernel under the tracer (ramp and warmup passes in front of them, strict ones behind)
import json
import re
try:
    line = json.loads([l for l in open(os.path.join(root, "bench_trace.json")) if l.startswith("{")][-1])
    ramp = int(re.match(r"(\d+) untimed", line["config"].get("clock_ramp", "0 untimed")).group(1))
    first, count = int(line["config"].get("cold_steps", 0)) + ramp + line["warmup"], line["steps"]
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "sample_counts_stream_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        timed = dur[first:first + count]
        print("== stream kernel under the tracer: %d dispatches, mean %.1f us; the %d of bench.py's timed region (dispatches %d..%d): "
              "mean %.1f us, min %.1f, max %.1f  (bench_trace.json: roofline.kernel_ms %.4f)"
              % (len(dur), sum(dur) / len(dur), len(timed), first, first + len(timed) - 1, sum(timed) / max(1, len(timed)),
                 min(timed), max(timed), line["roofline"]["kernel_ms"]))
except Exception as exc:                                      # older sets have no ramp information
    print("== (no per-dispatch summary: %r)" % (exc,))
for f in glob.glob(os.path.join(root, "*.json")):
    print("== %s ==" % os.path.basename(f))
    print(open(f).read().strip()[:2000])
allc = {}
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
                if "stream_kernel" in k:
                    allc[c] = sum(xs) / len(xs)

# ---- derived, stream kernel: the SQ's wave-cycle budget and the issue rate ----
need = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
        "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "GRBM_GUI_ACTIVE")
if all(c in allc for c in need):
    w = allc["SQ_WAVE_CYCLES"]
    parts = allc["SQ_WAIT_ANY"] + allc["SQ_WAIT_INST_ANY"] + allc["SQ_ACTIVE_INST_ANY"]
    print("== derived: k3::sample_counts_stream_kernel (per launch) ==")
    print("  wave cycles (quads) %.4g = waiting at s_waitcnt %.1f %% + waiting for issue %.1f %% + executing %.1f %% (sum %.1f %%)"
          % (w, 100 * allc["SQ_WAIT_ANY"] / w, 100 * allc["SQ_WAIT_INST_ANY"] / w, 100 * allc["SQ_ACTIVE_INST_ANY"] / w, 100 * parts / w))
    try:
        samples = float(line["config"]["cells_on_rank_0"]) * float(re.search(r"(\d+) genes", line["config"]["workload"]).group(1))
    except Exception:
        samples = 1e9
    instr = sum(allc[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD",
                                  "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"))
    cycles = allc["GRBM_GUI_ACTIVE"] / 8.0          # the counter sums the 8 XCDs
    simds = 1024.0
    print("  wave-level instructions %.4g (vector %.4g, scalar %.4g, LDS %.4g, branch %.4g, memory %.4g): %.1f lane-instructions per sample, "
          "%.3g per SIMD in %.4g cycles = %.2f cycles per instruction per SIMD"
          % (instr, allc["SQ_INSTS_VALU"], allc["SQ_INSTS_SALU"], allc["SQ_INSTS_LDS"], allc["SQ_INSTS_BRANCH"],
             allc["SQ_INSTS_SMEM"] + allc["SQ_INSTS_VMEM_RD"] + allc["SQ_INSTS_VMEM_WR"],
             instr * 64 / samples, instr / simds, cycles, cycles / (instr / simds)))
