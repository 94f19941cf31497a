#!/usr/bin/env python3
"""Where the first calls of a process spend their time (bench.py's ms_per_step_cold): context creation, the first
launch of the library's code object (a tiny call), the first full-size call (workspace allocation, the scan of the mean
tensor, idle clocks), the calls behind it.  usage (GPU box): tools/cold_start.py [C3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

t0 = time.perf_counter()
from prosstt_amd import device, workloads
torch.cuda.init()
torch.cuda.synchronize()
t_init = time.perf_counter() - t0


def timed(label, fn):
    torch.cuda.synchronize()
    a = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print("%-58s %9.3f ms" % (label, (time.perf_counter() - a) * 1e3))
    return r


print("%-58s %9.3f ms" % ("import + torch.cuda.init", t_init * 1e3))
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = timed("Context() (ctx_create: scratch, pinned mirror, occupancy query)", device.get_context)
w = workloads.build(cfg)
pt, br, sc, rows = w.plan()
G = w.tree.G
dm = w.tree.device_means()
dr, ds = ctx.tensor(rows, torch.int32), ctx.tensor(sc, torch.float64)
da, db = ctx.tensor(w.alpha, torch.float64), ctx.tensor(w.beta, torch.float64)
out = timed("torch.empty of the count matrix (%.0f MB)" % (len(rows) * G * 4 / 1e6),
            lambda: torch.empty((len(rows), G), dtype=torch.int32, device="cuda"))
token = w.tree.means_token()
small = torch.empty((64, G), dtype=torch.int32, device="cuda")
if os.environ.get("COLD_SKIP_SMALL") != "1":
    timed("first call of the process, 64 cells (code object, 1 MB workspace)",
          lambda: ctx.sample_counts(dm, dr[:64], ds[:64], da, db, seed=1, out=small, check_domain="deferred", means_token=token))
    timed("second small call", lambda: ctx.sample_counts(dm, dr[:64], ds[:64], da, db, seed=1, out=small, check_domain="deferred", means_token=token))
for i in range(6):
    timed("full-size call %d%s" % (i + 1, " (workspace grows to N*G/4 bytes; idle clocks)" if i == 0 else ""),
          lambda: ctx.sample_counts(dm, dr, ds, da, db, seed=2 + i, out=out, check_domain="deferred", means_token=token))
ctx.domain_status()
timed("torch.empty of 256 MB (what a hipMalloc of the list's size costs here)", lambda: torch.empty(256 << 20, dtype=torch.uint8, device="cuda"))
