"""Host only: what the host library's pool writes per second (int32 -> int64, uint16 -> int64, uint16 -> int32 of 1e9 counts
in page-locked memory), by thread count -- the ceiling of a host return whose transfer format is narrower than its type."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from prosstt_amd import _native  # noqa: E402

L = _native.load_host()
for name in ("prosstt_amd_host_widen_u16_i64", "prosstt_amd_host_widen_u16_i32"):
    getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32]
n = 1_000_000_000
pin = torch.cuda.is_available()
src32 = torch.ones(n, dtype=torch.int32, pin_memory=pin)
src16 = torch.ones(n, dtype=torch.int16, pin_memory=pin)
dst64 = torch.empty(n, dtype=torch.int64, pin_memory=pin)
dst32 = torch.empty(n, dtype=torch.int32, pin_memory=pin)
dst64.zero_(); dst32.zero_()
for threads in (2, 4, 8, 16, 32, 64):
    row = []
    for fn, s, d, out_bytes in ((L.prosstt_amd_host_widen_i32_i64, src32, dst64, 8), (L.prosstt_amd_host_widen_u16_i64, src16, dst64, 8),
                                (L.prosstt_amd_host_widen_u16_i32, src16, dst32, 4)):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            fn(s.data_ptr(), d.data_ptr(), n, threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row.append("%.1f ms (%.0f GB/s written)" % (best * 1e3, out_bytes * n / best / 1e9))
    print("%2d threads: i32->i64 %s | u16->i64 %s | u16->i32 %s" % (threads, *row))
