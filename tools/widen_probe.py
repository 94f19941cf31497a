"""Where the reference's int64 matrix is formed (device.WIDEN_ON): wall time of device.to_host for the C3 matrix,
int64 widened on the device (8 bytes per count over PCIe) against int32 over PCIe + widening on the host's threads,
with and without the row gather, for several thread counts.  Usage: python3 tools/widen_probe.py [cells] [genes]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from prosstt_amd import device  # noqa: E402


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = fn()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
        del x
    return best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    gen = torch.Generator(device="cuda").manual_seed(1)
    counts = (torch.rand((n, g), device="cuda", generator=gen) < 0.35).to(torch.int32) * \
        torch.randint(1, 300, (n, g), device="cuda", generator=gen, dtype=torch.int32)
    order = np.random.default_rng(0).permutation(n)
    print("host threads available: torch %d" % torch.get_num_threads())
    device.WIDEN_ON = "device"
    want = device.to_host(counts, np.int64)
    want_perm = device.to_host(counts, np.int64, row_order=order)
    print("device-widened: %.1f ms   with row gather %.1f ms   int32 %.1f ms" % (
        timed(lambda: device.to_host(counts, np.int64)), timed(lambda: device.to_host(counts, np.int64, row_order=order)),
        timed(lambda: device.to_host(counts, np.int32))))
    device.WIDEN_ON = "host"
    for threads in (4, 8, 12, 16, 20, 24, 32, 64):
        device.HOST_THREADS = threads
        got = device.to_host(counts, np.int64)
        same = np.array_equal(got, want)
        got = device.to_host(counts, np.int64, row_order=order)
        same_perm = np.array_equal(got, want_perm)
        del got
        print("host-widened, %2d threads: %.1f ms   with row gather %.1f ms   equal %s %s" % (
            threads, timed(lambda: device.to_host(counts, np.int64)),
            timed(lambda: device.to_host(counts, np.int64, row_order=order)), same, same_perm))
    for threads in (12, 16, 20):
        for chunk in (32 << 20, 64 << 20, 128 << 20, 512 << 20):
            device.HOST_THREADS = threads
            print("host-widened, %d threads, chunks of %d MB: %.1f ms" % (
                threads, chunk >> 20, timed(lambda: device.to_host(counts, np.int64, chunk_bytes=chunk))))


if __name__ == "__main__":
    main()
