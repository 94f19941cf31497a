"""
ORACLE -- test infrastructure only (see oracle/numpy_legacy.c).  ctypes binding
of the C restatement of numpy's legacy MT19937 negative-binomial chain.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libnumpy_legacy.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
        L = ctypes.CDLL(path)
        vp, i64, f64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
        L.npl_state_size.restype = ctypes.c_int
        L.npl_seed.argtypes = [vp, ctypes.c_uint32]
        L.npl_random_sample.argtypes = [vp, _f64p, i64]
        L.npl_standard_normal.argtypes = [vp, _f64p, i64]
        L.npl_standard_gamma.argtypes = [vp, f64, _f64p, i64]
        L.npl_poisson.argtypes = [vp, _f64p, _i64p, i64]
        L.npl_negative_binomial.argtypes = [vp, _f64p, _f64p, _i64p, i64]
        L.npl_draw_counts.argtypes = [vp, _f64p, i64, _i64p, _f64p, _f64p, _f64p, i64, _i64p]
        L.npl_draw_counts.restype = i64
        _LIB = L
    return _LIB


class LegacyState:
    """MT19937 + Gaussian cache, seeded like ``np.random.seed(int)``."""

    def __init__(self, seed):
        self._buf = ctypes.create_string_buffer(lib().npl_state_size())
        lib().npl_seed(self._buf, seed)

    def random_sample(self, n):
        out = np.empty(n)
        lib().npl_random_sample(self._buf, out, n)
        return out

    def standard_normal(self, n):
        out = np.empty(n)
        lib().npl_standard_normal(self._buf, out, n)
        return out

    def standard_gamma(self, shape, n):
        out = np.empty(n)
        lib().npl_standard_gamma(self._buf, shape, out, n)
        return out

    def poisson(self, lam):
        lam = np.ascontiguousarray(lam, np.float64)
        out = np.empty(lam.size, np.int64)
        lib().npl_poisson(self._buf, lam, out, lam.size)
        return out

    def negative_binomial(self, n, p):
        n = np.ascontiguousarray(n, np.float64)
        p = np.ascontiguousarray(p, np.float64)
        out = np.empty(n.size, np.int64)
        lib().npl_negative_binomial(self._buf, n, p, out, n.size)
        return out

    def draw_counts(self, means, row_of_cell, scaling, alpha, beta):
        means = np.ascontiguousarray(means, np.float64)
        row_of_cell = np.ascontiguousarray(row_of_cell, np.int64)
        out = np.empty((row_of_cell.size, means.shape[1]), np.int64)
        bad = lib().npl_draw_counts(self._buf, means, means.shape[1], row_of_cell,
                                    np.ascontiguousarray(scaling, np.float64),
                                    np.ascontiguousarray(alpha, np.float64),
                                    np.ascontiguousarray(beta, np.float64), row_of_cell.size, out)
        if bad:
            raise ValueError("Domain error in arguments.")
        return out
