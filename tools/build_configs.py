"""Build benchmark workloads through the product pipeline and report the lineage stage (attempts, seconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosstt_amd import workloads
for name in sys.argv[1:] or ["C3", "C4", "C5"]:
    t0 = time.perf_counter()
    w = workloads.build(name, verbose=True)
    print("%s built in %.1f s: %s" % (name, time.perf_counter() - t0, w.info), flush=True)
    del w
