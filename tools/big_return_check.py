"""The host returns at C4's size (200 000 x 20 000 = 4e9 counts, 32 GB as int64): int64, int32 and csr of the same plan and seed
agree with each other and with the device matrix; wall time of each (second call).  Usage: python3 tools/big_return_check.py [config]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from prosstt_amd import simulation as sim, workloads  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
work = workloads.build(cfg)
tree, n = work.tree, work.cfg["N"]


def call(out):
    best, x = None, None
    for _ in range(2):
        del x
        np.random.seed(work.cfg["seed"] + 1)
        t0 = time.perf_counter()
        x = sim.sample_density(tree, n, alpha=work.alpha, beta=work.beta, seed=77, out=out)[0]
        best = (time.perf_counter() - t0) * 1e3
    return x, best


x32, t32 = call("numpy32")
print("%s %d x %d: int32 %.0f ms, max count %d, zeros %.1f %%" % (cfg, n, tree.G, t32, int(x32.max()), 100.0 * float((x32[:2000] == 0).mean())))
x64, t64 = call("numpy")
same = all(np.array_equal(x64[lo:lo + 20000], x32[lo:lo + 20000]) for lo in range(0, n, 20000))
print("int64 %.0f ms, equal to int32: %s" % (t64, same))
del x64
s, ts = call("csr")
rows = np.random.default_rng(0).choice(n, 3000, replace=False)
print("csr %.0f ms, nnz %d, %d rows equal to int32: %s, every row's non-zero count: %s" % (
    ts, s.nnz, len(rows), np.array_equal(s[rows].toarray(), x32[rows]),
    np.array_equal(np.diff(s.indptr), np.count_nonzero(x32, axis=1))))
np.random.seed(work.cfg["seed"] + 1)
dev = sim.sample_density(tree, n, alpha=work.alpha, beta=work.beta, seed=77, out="torch")[0]
print("device matrix (rows put back) equal to int32 on 3000 rows: %s" % np.array_equal(dev.in_plan_order()[rows].cpu().numpy(), x32[rows]))
