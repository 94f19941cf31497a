"""
-m gpu: the C ABI called the way INTEGRATION.md's stub calls it -- raw ctypes, HOST pointers
(PROSSTT_AMD_HOST_INPUTS | HOST_OUTPUT), no torch tensors -- against the oracle model.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_sample_counts_and_params_with_host_buffers():
    from prosstt_amd import _native
    from oracle import nb_model
    L = _native.load()
    ctx = ctypes.c_void_p()
    assert L.prosstt_amd_ctx_create(0, None, ctypes.byref(ctx)) == 0
    try:
        rng = np.random.default_rng(4)
        rows, G, N = 23, 777, 130
        means = np.exp(rng.normal(0.7, 1.4, (rows, G))).astype(np.float32)
        means[:, ::50] *= 300                       # some gamma-Poisson samples
        roc = rng.integers(0, rows, N).astype(np.int32)
        sc = np.exp(rng.normal(0, 0.7, N))
        al = np.exp(rng.normal(np.log(0.2), 0.4, G))
        be = np.exp(rng.normal(0, 0.4, G)) + 1
        idx = rng.permutation(10 * N)[:N].astype(np.int64)
        out = np.full((N, G + 3), -7, np.int32)     # ld_out > G: the padding must stay untouched
        flags = _native.HOST_INPUTS | _native.HOST_OUTPUT | _native.CHECK_DOMAIN
        rc = L.prosstt_amd_sample_counts(ctx, _ptr(means), rows, G, _ptr(roc), _ptr(sc), _ptr(al), _ptr(be), N,
                                         ctypes.c_uint64(99), ctypes.c_uint64(0), _ptr(idx), _ptr(out), G + 3, flags)
        assert rc == 0, L.prosstt_amd_last_error()
        want = nb_model.sample_counts(means, roc, sc, al, be, 99, cell_index=idx)
        np.testing.assert_array_equal(out[:, :G], want)
        assert (out[:, G:] == -7).all()
        # staged output copies whole rows of ld_out elements; only the first G of each are defined
        mu = np.empty((N, G), np.float32)
        p = np.empty((N, G), np.float32)
        path = np.empty((N, G), np.int32)
        rc = L.prosstt_amd_nb_params(ctx, _ptr(means), rows, G, _ptr(roc), _ptr(sc), _ptr(al), _ptr(be), N,
                                     _ptr(mu), _ptr(p), None, _ptr(path), _native.HOST_INPUTS | _native.HOST_OUTPUT)
        assert rc == 0
        mmu, mp, _, mpath = nb_model.nb_params(means, roc, sc, al, be)
        np.testing.assert_array_equal(mu, mmu)
        np.testing.assert_array_equal(p, mp)
        np.testing.assert_array_equal(path, mpath)
        # error reporting: bad row index (checked on the host for host inputs), zero mean -> EDOMAIN
        bad = roc.copy()
        bad[5] = rows
        rc = L.prosstt_amd_sample_counts(ctx, _ptr(means), rows, G, _ptr(bad), _ptr(sc), _ptr(al), _ptr(be), N,
                                         ctypes.c_uint64(1), ctypes.c_uint64(0), None, _ptr(out), G + 3, flags)
        assert rc == _native.EINVAL and b"row_of_cell" in L.prosstt_amd_last_error()
        means[3, 3] = 0.0
        roc[0] = 3
        rc = L.prosstt_amd_sample_counts(ctx, _ptr(means), rows, G, _ptr(roc), _ptr(sc), _ptr(al), _ptr(be), N,
                                         ctypes.c_uint64(1), ctypes.c_uint64(0), None, _ptr(out), G + 3, flags)
        assert rc == _native.EDOMAIN
        rc = L.prosstt_amd_sample_counts(ctx, _ptr(means), rows, G, _ptr(roc), _ptr(sc), _ptr(al), _ptr(be), N,
                                         ctypes.c_uint64(1), ctypes.c_uint64(0), None, _ptr(out), G - 1, flags)
        assert rc == _native.EINVAL
    finally:
        L.prosstt_amd_ctx_destroy(ctx)


def test_both_kernels_agree_on_a_heavy_workload():
    """Streaming + heavy kernels vs the tile-per-block kernel on inputs that are mostly gamma-Poisson."""
    import os
    import torch
    from prosstt_amd import device
    ctx = device.get_context()
    rng = np.random.default_rng(8)
    rows, G, N = 40, 1300, 700
    means = np.exp(rng.normal(4.0, 1.5, (rows, G))).astype(np.float32)
    roc = rng.integers(0, rows, N).astype(np.int32)
    sc = np.exp(rng.normal(0, 0.7, N))
    al = np.exp(rng.normal(np.log(0.6), 0.8, G))
    be = np.exp(rng.normal(0, 0.6, G)) + 1
    a = ctx.sample_counts(means, roc, sc, al, be, seed=5)
    from oracle import nb_model
    np.testing.assert_array_equal(a.cpu().numpy(), nb_model.sample_counts(means, roc, sc, al, be, 5))
    _, _, _, path = ctx.nb_params(means, roc, sc, al, be)
    assert float((path == 2).float().mean()) > 0.5
