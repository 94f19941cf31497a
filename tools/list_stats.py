#!/usr/bin/env python3
"""Composition of the list the streaming kernel leaves to K3h on a benchmark config (GPU box).
usage: tools/list_stats.py [C3|C4|C5] [cells]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from prosstt_amd import device, workloads
    from oracle import nb_model
    nb_model.install_hw_tables_from_device()
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    ctx = device.get_context()
    work = workloads.build(cfg)
    N = int(sys.argv[2]) if len(sys.argv) > 2 else work.cfg["N"]
    pt, br, sc, rows = work.plan(N)
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=424242)
    cells, genes, total, over = ctx.last_list(cap=1 << 25)
    path, count = nb_model.sample_selected(means.cpu().numpy(), rows, sc, work.alpha, work.beta, 424242, cells, genes)
    heavy = path == 2
    big = (path == 1) & (count > 248)
    rest = (path == 1) & ~big
    samples = N * work.tree.G
    print("%s %d x %d: %d listed (%.3f %% of the samples, overflowed %s): gamma-Poisson %d, walks past k = 248: %d, "
          "unfinished at the end of their strip %d (mean count %.1f, max %d); mean count of the matrix %.2f"
          % (cfg, N, work.tree.G, total, 100.0 * total / samples, over, heavy.sum(), big.sum(), rest.sum(),
             count[rest].mean() if rest.any() else 0, count[rest].max() if rest.any() else 0,
             float(X[:4096].double().mean())))
    q = np.percentile(count[rest], [50, 90, 99]) if rest.any() else [0, 0, 0]
    print("   unfinished walks: count percentiles 50/90/99 = %d/%d/%d" % tuple(q))
    if heavy.any():
        # parameters of the gamma-Poisson class: which branches of K3h its samples take
        hc, hg = cells[heavy], genes[heavy]
        M = means.cpu().numpy()
        m = (M[np.asarray(rows)[hc], hg].astype(np.float64) * np.asarray(sc, dtype=np.float64)[hc])
        theta = np.asarray(work.alpha, dtype=np.float64)[hg] * m + np.asarray(work.beta, dtype=np.float64)[hg] - 1.0
        r = m / theta
        cnt = count[heavy]
        print("   gamma-Poisson class: theta > 24: %.1f %%; shape r = m/theta < 1: %.1f %%; m percentiles 10/50/90 = %.1f/%.1f/%.1f; "
              "theta 10/50/90 = %.1f/%.1f/%.1f; counts: 0: %.1f %%, < 10: %.1f %%, mean %.1f"
              % (100.0 * (theta > 24).mean(), 100.0 * (r < 1).mean(), *np.percentile(m, [10, 50, 90]), *np.percentile(theta, [10, 50, 90]),
                 100.0 * (cnt == 0).mean(), 100.0 * (cnt < 10).mean(), cnt.mean()))


if __name__ == "__main__":
    main()
