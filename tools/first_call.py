"""What a one-shot script pays: the FIRST sample_density call of a process (C3, int64 ndarray) and the two after it, with the
result in ordinary memory (numpy blocks on transparent huge pages, recycled once the caller has dropped the result: the default) and
in page-locked memory from torch's caching host allocator (PROSSTT_AMD_RESULT_MEMORY=pinned).  One box: pageable 159 - 423 / 47 - 51 / 40 - 42 ms;
pinned 724 / 80 / 80 ms (other boxes: 564 - 958 / 54 / 59 ms).  Usage: python3 tools/first_call.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
t_import = time.perf_counter()
from prosstt_amd import device, simulation as sim, workloads  # noqa: E402

work = workloads.build("C3")
tree, n = work.tree, work.cfg["N"]
print("result memory: %s, %d host threads" % (device.RESULT_MEMORY, device.HOST_THREADS))
for i in range(3):
    np.random.seed(work.cfg["seed"] + 1)
    t0 = time.perf_counter()
    x = sim.sample_density(tree, n, alpha=work.alpha, beta=work.beta)[0]
    dt = (time.perf_counter() - t0) * 1e3
    print("call %d: %.1f ms   (sum of row 0: %d)" % (i + 1, dt, int(x[0].sum())))
    del x
