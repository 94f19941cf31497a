"""End to end of the drop-in call (sample_density -> int64 / int32 ndarray) on C3 by wire format (device.WIRE) and size of the host
library's widening pool, beside the device-widened copies, the sparse and the uint16 return.  Usage: python3 tools/e2e_threads.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from prosstt_amd import device, simulation as sim, workloads  # noqa: E402

work = workloads.build("C3")
tree, n = work.tree, work.cfg["N"]


def call(out):
    best = None
    for _ in range(3):
        np.random.seed(work.cfg["seed"] + 1)
        t0 = time.perf_counter()
        x = sim.sample_density(tree, n, alpha=work.alpha, beta=work.beta, out=out)[0]
        dt = (time.perf_counter() - t0) * 1e3
        del x
        best = dt if best is None else min(best, dt)
    return best


for wire in ("u8", "u16", "i32"):
    device.WIRE = wire
    for threads in (8, 16, 32):
        device.HOST_THREADS = threads
        print("wire %s, %2d host threads: int64 %.1f ms   int32 %.1f ms   csr %.1f ms" % (wire, threads, call("numpy"), call("numpy32"), call("csr")))
device.WIDEN_ON = "device"
print("widened on the device / copied as it lies: int64 %.1f ms   int32 %.1f ms   csr %.1f ms" % (call("numpy"), call("numpy32"), call("csr")))
