"""End-to-end (PCIe-inclusive) time of the drop-in API on config C3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosstt_amd import simulation as sim, workloads
w = workloads.build("C3")
for out in ("torch", "numpy"):
    for rep in range(2):
        np.random.seed(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        X, pt, br, sc = sim.sample_density(w.tree, 50000, alpha=w.alpha, beta=w.beta, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("sample_density(50k cells x 20k genes, out=%s): %.3f s  (%.2e cells*genes/s) %s %s" % (out, dt, 1e9 / dt, X.dtype, tuple(X.shape)))
        del X
