"""Host only: first touch of an 8 GB result by the host library's pool (uint16 -> int64 of 1e9 counts) into (a) a fresh numpy
array (numpy asks for transparent huge pages on large allocations), (b) a fresh torch CPU tensor, (c) page-locked memory
(allocation time printed separately), and the second touch of each."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from prosstt_amd import _native  # noqa: E402

print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(),
      "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
L = _native.load_host()
n = 1_000_000_000
src = torch.ones(n, dtype=torch.int16, pin_memory=torch.cuda.is_available())


def touch(ptr, threads=16):
    t0 = time.perf_counter()
    L.prosstt_amd_host_widen_u16_i64(src.data_ptr(), ptr, n, threads)
    return (time.perf_counter() - t0) * 1e3


for rep in range(2):
    t0 = time.perf_counter(); a = np.empty(n, np.int64); ta = (time.perf_counter() - t0) * 1e3
    print("numpy: alloc %.1f ms, first touch %.1f ms, second %.1f ms" % (ta, touch(a.ctypes.data), touch(a.ctypes.data)))
    del a
    t0 = time.perf_counter(); b = torch.empty(n, dtype=torch.int64); tb = (time.perf_counter() - t0) * 1e3
    print("torch: alloc %.1f ms, first touch %.1f ms, second %.1f ms" % (tb, touch(b.data_ptr()), touch(b.data_ptr())))
    del b
if torch.cuda.is_available():
    for rep in range(2):
        t0 = time.perf_counter(); c = torch.empty(n, dtype=torch.int64, pin_memory=True); tc = (time.perf_counter() - t0) * 1e3
        print("pinned: alloc %.1f ms, first touch %.1f ms, second %.1f ms" % (tc, touch(c.data_ptr()), touch(c.data_ptr())))
        del c
