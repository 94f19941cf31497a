#!/usr/bin/env python3
"""Finer than tools/cold_start.py: which first launch of the process pays for the code object?  (GPU box)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
t0 = time.perf_counter(); torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print("torch.cuda.init + first tensor            %9.3f ms" % ((time.perf_counter() - t0) * 1e3))
from prosstt_amd import _native, device
def timed(label, fn):
    torch.cuda.synchronize(); a = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-42s %9.3f ms" % (label, (time.perf_counter() - a) * 1e3)); return r
lib = timed("dlopen of the library (_native.load)", _native.load)
ctx = timed("Context()", device.get_context)
G = 512
means = torch.rand((10, G), device="cuda") + 0.1
rows = torch.zeros(64, dtype=torch.int32, device="cuda"); sc = torch.ones(64, dtype=torch.float64, device="cuda")
al = torch.full((G,), 0.2, dtype=torch.float64, device="cuda"); be = torch.full((G,), 2.0, dtype=torch.float64, device="cuda")
timed("nb_params, 64 x 512 (first launch of the process)", lambda: ctx.nb_params(means, rows, sc, al, be))
timed("nb_params again", lambda: ctx.nb_params(means, rows, sc, al, be))
timed("sample_counts unchecked, 64 x 512", lambda: ctx.sample_counts(means, rows, sc, al, be, seed=1, check_domain=False))
timed("sample_counts unchecked again", lambda: ctx.sample_counts(means, rows, sc, al, be, seed=1, check_domain=False))
timed("sample_counts deferred check", lambda: ctx.sample_counts(means, rows, sc, al, be, seed=1, check_domain="deferred"))
timed("sample_counts deferred again", lambda: ctx.sample_counts(means, rows, sc, al, be, seed=1, check_domain="deferred"))
timed("domain_status", ctx.domain_status)
timed("sample_counts time_kernel (events)", lambda: ctx.sample_counts(means, rows, sc, al, be, seed=1, check_domain=False, time_kernel=True))
timed("last_kernel_ms", ctx.last_kernel_ms)
big = timed("torch.empty 1 GB", lambda: torch.empty(1 << 30, dtype=torch.uint8, device="cuda"))
