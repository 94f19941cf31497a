"""Kernel-only timing of K3 on a benchmark config (default C3) -- used for A/B and ablations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosstt_amd import device, workloads
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = device.get_context()
w = workloads.build(cfg)
pt, br, sc, rows = w.plan(int(os.environ["KBENCH_CELLS"]) if "KBENCH_CELLS" in os.environ else (125000 if cfg == "C5" else None))
sc = sc * float(os.environ.get('KBENCH_SCALE', '1'))
if os.environ.get("KBENCH_SORT") == "1":
    o = np.argsort(rows, kind="stable"); rows, sc = np.asarray(rows)[o], np.asarray(sc)[o]
G = w.tree.G
dm = w.tree.device_means(); dr = ctx.tensor(rows, torch.int32); ds = ctx.tensor(sc, torch.float64)
da = ctx.tensor(w.alpha, torch.float64); db = ctx.tensor(w.beta, torch.float64)
out = torch.empty((len(rows), G), dtype=torch.int32, device='cuda')
ts, calls = [], []
for i in range(int(os.environ.get('KBENCH_ITERS', '16'))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.sample_counts(dm, dr, ds, da, db, seed=i, out=out, check_domain=False, time_kernel=True)
    e1.record(); torch.cuda.synchronize()
    ts.append(ctx.last_kernel_ms()); calls.append(e0.elapsed_time(e1))
n = len(rows) * G
ms = float(np.median(ts[1:]))
print("%-34s %s: call median %.3f ms | kernel median %.3f min %.3f ms  %.1f G samples/s  %.1f %% of 8 TB/s  (sum %d)" % (
    os.environ.get("PROSSTT_AMD_LIB", "default")[-34:], cfg, float(np.median(calls[1:])), ms, min(ts[1:]), n / ms / 1e6,
    n * 4.0325 / (ms * 1e-3) / 8e12 * 100, int(out.sum())))
