"""
ORACLE -- test infrastructure only (see oracle/nb_model.c).  ctypes binding of
the C model of the device count sampler.  Never imported by prosstt_amd.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libprnb_model.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.prnb_philox.argtypes = [_u32p, _u32p, _u32p]
        L.prnb_philox_rounds.argtypes = [ctypes.c_int, _u32p, _u32p, _u32p]
        L.prnb_philox_rounds.restype = None
        L.prnb_count_rounds.restype = ctypes.c_int
        L.prnb_math.argtypes = [ctypes.c_int, _f32p, _f32p, ctypes.c_int64]
        L.prnb_sample_counts.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _i32p, _f64p, _f64p,
                                         _f64p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_void_p, _i32p, ctypes.c_int64]
        L.prnb_nb_params.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _i32p, _f64p, _f64p, _f64p,
                                     ctypes.c_int64, _f32p, _f32p, _f32p, _i32p]
        _i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        L.prnb_walk_detail.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _i32p, _f64p, _f64p, _f64p, ctypes.c_uint64,
                                       ctypes.c_uint64, ctypes.c_void_p, _i64p, _i32p, ctypes.c_int64, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_float, _i32p, _i32p, _f32p, _f32p, _f32p]
        L.prnb_walk_detail.restype = None
        L.prnb_sample_iid.argtypes = [ctypes.c_float, ctypes.c_double, ctypes.c_double, ctypes.c_uint64,
                                      ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64, _i32p]
        L.prnb_lineage_walk.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, _f64p]
        L.prnb_lineage_walk.restype = None
        for f in (L.prnb_philox, L.prnb_math, L.prnb_sample_counts, L.prnb_nb_params, L.prnb_sample_iid):
            f.restype = None
        _LIB = L
    return _LIB


def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().prnb_philox(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def philox_rounds(rounds, ctr, key):
    out = np.zeros(4, np.uint32)
    lib().prnb_philox_rounds(rounds, np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def count_rounds():
    """Rounds of the Philox4x32 generator behind the count sampler."""
    return lib().prnb_count_rounds()


MATH = dict(rcp=0, log=1, log1p=2, exp=3, cos2pi=4, unif=5, log1pmx=6)


def math(which, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().prnb_math(MATH[which], x, y, x.size)
    return y


def sample_counts(means, row_of_cell, scaling, alpha, beta, seed, cell_offset=0, cell_index=None):
    means = np.ascontiguousarray(means, np.float32)
    rows, G = means.shape
    row_of_cell = np.ascontiguousarray(row_of_cell, np.int32)
    N = row_of_cell.size
    out = np.empty((N, G), np.int32)
    if cell_index is not None:
        cell_index = np.ascontiguousarray(cell_index, np.int64)
    lib().prnb_sample_counts(means, rows, G, row_of_cell, np.ascontiguousarray(scaling, np.float64),
                             np.ascontiguousarray(alpha, np.float64),
                             np.ascontiguousarray(beta, np.float64), N, seed, cell_offset,
                             cell_index.ctypes.data if cell_index is not None else None, out, G)
    return out


def walk_detail(means, row_of_cell, scaling, alpha, beta, seed, cells, genes, margins, cell_offset=0, cell_index=None):
    """(path, count, t2, close, tail) of selected samples: how close the exact walk comes to a decision the
    device's hardware-math evaluation could get wrong (see prnb_walk_detail).  margins = the device's
    (margin0, margin per unit of t2, margin per term), in units of 2^-32."""
    means = np.ascontiguousarray(means, np.float32)
    rows, G = means.shape
    cells = np.ascontiguousarray(cells, np.int64)
    genes = np.ascontiguousarray(genes, np.int32)
    n = cells.size
    path, count = np.empty(n, np.int32), np.empty(n, np.int32)
    t2, close, tail = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    if cell_index is not None:
        cell_index = np.ascontiguousarray(cell_index, np.int64)
    lib().prnb_walk_detail(means, rows, G, np.ascontiguousarray(row_of_cell, np.int32),
                           np.ascontiguousarray(scaling, np.float64), np.ascontiguousarray(alpha, np.float64),
                           np.ascontiguousarray(beta, np.float64), seed, cell_offset,
                           cell_index.ctypes.data if cell_index is not None else None, cells, genes, n,
                           margins[0], margins[1], margins[2], path, count, t2, close, tail)
    return path, count, t2, close, tail


def nb_params(means, row_of_cell, scaling, alpha, beta):
    means = np.ascontiguousarray(means, np.float32)
    rows, G = means.shape
    row_of_cell = np.ascontiguousarray(row_of_cell, np.int32)
    N = row_of_cell.size
    mu = np.empty((N, G), np.float32)
    p = np.empty((N, G), np.float32)
    r = np.empty((N, G), np.float32)
    path = np.empty((N, G), np.int32)
    lib().prnb_nb_params(means, rows, G, row_of_cell, np.ascontiguousarray(scaling, np.float64),
                         np.ascontiguousarray(alpha, np.float64),
                         np.ascontiguousarray(beta, np.float64), N, mu, p, r, path)
    return mu, p, r, path


def sample_iid(m, a, b, n, seed=0, first_cell=0, gene=0):
    out = np.empty(n, np.int32)
    lib().prnb_sample_iid(m, a, b, seed, first_cell, gene, n, out)
    return out


def lineage_walk(seed, sid, T, K):
    out = np.empty((T, K), np.float64)
    lib().prnb_lineage_walk(seed, sid, T, K, out)
    return out
