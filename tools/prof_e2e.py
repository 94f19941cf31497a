import sys, time, cProfile, pstats
sys.path.insert(0, ".")
import numpy as np
from prosstt_amd import device, simulation as sim, workloads
work = workloads.build("C3")
tree, n = work.tree, work.cfg["N"]
for out in ("numpy32", "numpy"):
    for _ in range(2):
        np.random.seed(1); x = sim.sample_density(tree, n, alpha=work.alpha, beta=work.beta, out=out)[0]; del x
    np.random.seed(1)
    pr = cProfile.Profile(); pr.enable()
    x = sim.sample_density(tree, n, alpha=work.alpha, beta=work.beta, out=out)[0]
    pr.disable(); del x
    print("====", out)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
