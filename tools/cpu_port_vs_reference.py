#!/usr/bin/env python3
"""
Build container only (needs /root/reference): is bench.py's CPU baseline -- the oracle's
``draw_counts_as_reference`` -- the reference's ``simulation.draw_counts`` in cost and in output?

Same tree, plan and parameters for both; the two draw from the same numpy seed, so the matrices must be
identical; wall times are interleaved (3 rounds) and must agree within 10 %.

    python tools/cpu_port_vs_reference.py [cells] [genes]
"""
import os
import sys
import time
import types
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("PROSSTT_REFERENCE", "/root/reference"))
sys.modules["newick"] = types.ModuleType("newick")
warnings.filterwarnings("ignore")

import numpy as np                                   # noqa: E402
from prosstt import tree as rtree                    # noqa: E402
from prosstt import simulation as rsim               # noqa: E402
from oracle import ref_numpy                         # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 600
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
topology = [[0, 1], [0, 2], [2, 3], [2, 4], [4, 5], [4, 6], [6, 7]]           # 8 branches, like C3
time_of = {b: 50 for b in range(8)}
rng = np.random.default_rng(1)
means = {b: np.exp(rng.normal(0.3, 1.2, (50, G))) for b in range(8)}
t = rtree.Tree(topology=topology, time=time_of, num_branches=8, branch_points=3, modules=25, G=G)
t.means = means
o = ref_numpy.RefTree(topology, time_of, modules=25, G=G)
o.means = means
bt = t.branch_times()
br = rng.integers(0, 8, cells)
pt = np.array([bt[b][0] + rng.integers(0, 50) for b in br])
sc = np.exp(rng.normal(0, 0.7, cells))
alpha = np.exp(rng.normal(np.log(0.2), np.log(1.5), G))
beta = np.exp(rng.normal(0, np.log(1.5), G)) + 1

best = {"reference": 1e9, "port": 1e9, "port (hoisted)": 1e9}
for rnd in range(3):
    for name, fn, tree in (("reference", rsim.draw_counts, t), ("port", ref_numpy.draw_counts_as_reference, o),
                           ("port (hoisted)", ref_numpy.draw_counts, o)):
        np.random.seed(5)
        t0 = time.perf_counter()
        X = fn(tree, pt, list(br), sc, alpha, beta)
        best[name] = min(best[name], time.perf_counter() - t0)
        if name == "reference":
            want = X
        else:
            assert np.array_equal(X, want), name
for name, dt in best.items():
    print("%-16s %7.3f s  %6.2f M cells*genes/s  (%.2f x the reference's time)"
          % (name, dt, cells * G / dt / 1e6, dt / best["reference"]))
ratio = best["port"] / best["reference"]
print("port / reference = %.3f -> %s" % (ratio, "within 10 %" if abs(ratio - 1) <= 0.10 else "NOT within 10 %"))
