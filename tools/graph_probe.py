"""What one hipGraph launch per call would buy on a small problem (C2: 5 000 x 5 000): the three kernels of a
sample_counts call (prep, stream kernel, K3h) captured once on a side stream (torch.cuda.CUDAGraph) and replayed,
against the same call launched kernel by kernel.  Device inputs, unchecked call (nothing synchronises inside).
usage: python3 tools/graph_probe.py [C2|C3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosstt_amd import device, workloads

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
w = workloads.build(cfg)
pt, br, sc, rows = w.plan(None)
o = np.argsort(rows, kind="stable"); rows, sc = np.asarray(rows)[o], np.asarray(sc)[o]
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ctx = device.get_context()
    G = w.tree.G
    dm = w.tree.device_means(); dr = ctx.tensor(rows, torch.int32); ds = ctx.tensor(sc, torch.float64)
    da = ctx.tensor(w.alpha, torch.float64); db = ctx.tensor(w.beta, torch.float64)
    out = torch.empty((len(rows), G), dtype=torch.int32, device="cuda")
    def call(seed):
        ctx.sample_counts(dm, dr, ds, da, db, seed=seed, out=out, check_domain=False)
    for i in range(5):
        call(i)
    torch.cuda.synchronize()
    want = out.clone()
    def timed(fn, n=200):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    plain = min(timed(lambda: call(4)) for _ in range(5))
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            call(4)
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(out, want))
        graph = min(timed(g.replay) for _ in range(5))
        print("%s: call kernel by kernel %.1f us | one graph launch %.1f us (%+.1f %%) | same counts: %s" % (cfg, plain, graph, (graph / plain - 1) * 100, same))
    except Exception as exc:
        print("%s: call kernel by kernel %.1f us | capture failed: %r" % (cfg, plain, exc))
