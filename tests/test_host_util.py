"""
CPU: libprosstt_amd_host.so (include/prosstt_amd_host.h) -- the int32 -> int64 widening the int64 host return runs on the
host's threads (prosstt_amd/device.py, _to_host_widened): equal to numpy's astype for every length around the vector
width, every alignment of the destination, any thread count; NULL refused; usable from two threads at once and from a
forked child of a process whose pool already runs.
"""
import ctypes
import multiprocessing
import os
import re
import threading

import numpy as np
import pytest

from conftest import ROOT


def lib():
    from prosstt_amd import _native
    if not os.path.exists(_native.HOST_LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "prosstt_amd", "csrc", "host"), "all"])
    return _native.load_host()


def widen(L, x, threads, offset=0):
    y = np.full(x.size + offset + 5, -7, dtype=np.int64)
    rc = L.prosstt_amd_host_widen_i32_i64(x.ctypes.data, y.ctypes.data + 8 * offset, x.size, threads)
    assert rc == 0
    assert np.all(y[:offset] == -7) and np.all(y[offset + x.size:] == -7), "wrote outside its range"
    return y[offset:offset + x.size]


def test_header_and_library_agree():
    from prosstt_amd import _native
    L = lib()
    text = open(os.path.join(ROOT, "include", "prosstt_amd_host.h")).read()
    declared = sorted(set(re.findall(r"\b(prosstt_amd_host_\w+)\s*\(", text)))
    assert declared == sorted(_native.HOST_SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.prosstt_amd_host_has_avx2() in (0, 1)


def test_widening_equals_astype_for_every_length_alignment_and_thread_count():
    L = lib()
    rng = np.random.default_rng(3)
    edge = np.array([np.iinfo(np.int32).min, -1, 0, 1, np.iinfo(np.int32).max], dtype=np.int32)
    for n in (0, 1, 3, 4, 15, 16, 17, 63, 64, 65, 1000, (1 << 14) + 1, 100003, 5_000_001):
        x = rng.integers(-2 ** 31, 2 ** 31, size=n, dtype=np.int64).astype(np.int32)
        x[:min(n, 5)] = edge[:min(n, 5)]
        for threads in (1, 2, 7, 64, 1000, 0, -3):
            for offset in (0, 1, 2, 3):
                got = widen(L, x, threads, offset)
                assert got.dtype == np.int64 and np.array_equal(got, x.astype(np.int64)), (n, threads, offset)
    assert L.prosstt_amd_host_widen_i32_i64(None, None, 0, 4) == 0
    assert L.prosstt_amd_host_widen_i32_i64(None, None, 10, 4) == -1


@pytest.mark.parametrize("wire", [np.uint16, np.uint8])
@pytest.mark.parametrize("out_dtype", [np.int64, np.int32])
def test_widening_from_the_narrow_wire_formats(out_dtype, wire):
    L = lib()
    fn = getattr(L, "prosstt_amd_host_widen_%s_%s" % ("u16" if wire == np.uint16 else "u8", "i64" if out_dtype == np.int64 else "i32"))
    rng = np.random.default_rng(5)
    top = int(np.iinfo(wire).max)
    for n in (0, 1, 7, 8, 15, 16, 17, 31, 33, 1000, (1 << 14) + 3, 2_000_003):
        x = rng.integers(0, top + 1, size=n, dtype=np.int64).astype(wire)
        x[:min(n, 2)] = np.array([top, 0], dtype=wire)[:min(n, 2)]
        for threads in (1, 3, 64):
            for offset in (0, 1, 3, 5):
                y = np.full(n + offset + 9, -7, dtype=out_dtype)
                assert fn(x.ctypes.data, y.ctypes.data + y.itemsize * offset, n, threads) == 0
                assert np.all(y[:offset] == -7) and np.all(y[offset + n:] == -7), "wrote outside its range"
                assert np.array_equal(y[offset:offset + n], x.astype(out_dtype)), (n, threads, offset)
    assert fn(None, None, 10, 2) == -1


@pytest.mark.parametrize("out_dtype", [np.int64, np.int32])
def test_scatter_of_the_counts_that_did_not_fit_the_wire(out_dtype):
    L = lib()
    rng = np.random.default_rng(6)
    size = 3_000_017
    for count in (0, 1, 100, (1 << 14) + 5, 700_001):
        for threads in (1, 4, 64):
            dst = rng.integers(0, 200, size=size).astype(out_dtype)
            pos = np.sort(rng.permutation(size)[:count]).astype(np.int64)
            val = rng.integers(-2 ** 31, 2 ** 31, size=count, dtype=np.int64).astype(np.int32)
            want = dst.copy()
            want[pos] = val
            assert L.prosstt_amd_host_scatter_i32(dst.ctypes.data, dst.itemsize, pos.ctypes.data, val.ctypes.data, count, threads) == 0
            assert np.array_equal(dst, want), (count, threads)
    one = np.zeros(4, out_dtype)
    assert L.prosstt_amd_host_scatter_i32(one.ctypes.data, 2, one.ctypes.data, one.ctypes.data, 1, 1) == -1       # an item size it does not write
    assert L.prosstt_amd_host_scatter_i32(one.ctypes.data, one.itemsize, None, None, 1, 1) == -1


def test_two_callers_at_once():
    L = lib()
    rng = np.random.default_rng(4)
    xs = [rng.integers(-2 ** 31, 2 ** 31, size=3_000_000 + i, dtype=np.int64).astype(np.int32) for i in range(2)]
    bad = []

    def work(x):
        for _ in range(6):
            if not np.array_equal(widen(L, x, 5), x.astype(np.int64)):
                bad.append(1)

    ts = [threading.Thread(target=work, args=(x,)) for x in xs]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not bad and not any(t.is_alive() for t in ts)


def _child(q):
    L = lib()
    x = np.arange(-500_000, 500_000, dtype=np.int32)
    q.put(bool(np.array_equal(widen(L, x, 6), x.astype(np.int64))))


def test_a_forked_child_gets_its_own_pool():
    L = lib()
    x = np.arange(2_000_000, dtype=np.int32)
    assert np.array_equal(widen(L, x, 6), x.astype(np.int64))         # the parent's pool runs
    ctx = multiprocessing.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_child, args=(q,))
    p.start()
    p.join(60)
    if p.is_alive():
        p.kill()
        pytest.fail("the child hung: it waited for the parent's worker threads")
    assert p.exitcode == 0 and q.get(timeout=5) is True
