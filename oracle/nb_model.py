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
        L.prnb_set_count_rounds.argtypes = [ctypes.c_int]
        L.prnb_set_count_rounds.restype = None
        L.prnb_math.argtypes = [ctypes.c_int, _f32p, _f32p, ctypes.c_int64]
        L.prnb_sample_counts.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _i32p, _f64p, _f64p,
                                         _f64p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_void_p, _i32p, ctypes.c_int64]
        L.prnb_nb_params.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _i32p, _f64p, _f64p, _f64p,
                                     ctypes.c_int64, _f32p, _f32p, _f32p, _i32p]
        L.prnb_set_hw_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
        L.prnb_set_hw_tables.restype = None
        L.prnb_hw_mode.restype = ctypes.c_int
        L.prnb_set_hw_query.argtypes = [ctypes.c_void_p]
        L.prnb_set_hw_query.restype = None
        L.prnb_set_hw_query_standins.restype = None
        L.prnb_query_stats.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.prnb_query_stats.restype = None
        L.prnb_hw_math.argtypes = [ctypes.c_int, _f32p, _f32p, ctypes.c_int64]
        L.prnb_hw_math.restype = None
        _i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        L.prnb_sample_selected.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _i32p, _f64p, _f64p, _f64p, ctypes.c_uint64,
                                           ctypes.c_uint64, ctypes.c_void_p, _i64p, _i32p, ctypes.c_int64, _i32p, _i32p]
        L.prnb_sample_selected.restype = None
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


class philox_rounds_for_draws:
    """``with philox_rounds_for_draws(10): ...`` -- the model's draws with another round count (the joint-law tests hold
    the 10-round matrix beside the 7-round definition's); the definition's count is restored on exit."""

    def __init__(self, rounds):
        self.rounds = int(rounds)

    def __enter__(self):
        lib().prnb_set_count_rounds(self.rounds)

    def __exit__(self, *exc):
        lib().prnb_set_count_rounds(0)


MATH = dict(rcp=0, log=1, log1p=2, exp=3, cos2pi=4, unif=5, log1pmx=6)

# ---- the hardware-function tables of PRNB-6 (see nb_model.c: hw_rcp / hw_log2 / hw_exp2neg) -------------------
# bit-pattern ranges the tables cover: every argument the sampler can present lies inside
ONE, TWO, THIRTY_TWO, TWO_M24 = 0x3F800000, 0x40000000, 0x42000000, 0x33800000
HW_RANGES = dict(rcp=(ONE, TWO - ONE),                       # mantissas: v_rcp_f32 over [1, 2)
                 log2=(ONE, THIRTY_TWO - ONE),               # v_log_f32 over [1, 32)
                 exp2neg=(TWO_M24, THIRTY_TWO - TWO_M24))    # v_exp_f32(-x) over [2^-24, 32)
_HW_TABLES = None      # the installed numpy arrays (kept alive here)
_HW_QUERY = None       # the installed ctypes callback (kept alive here)
# op codes of the query (= nb_model.c's HWOP_ enum = prosstt_amd_hw_math_at's op)
HW_OPS = dict(rcp=0, log2=1, exp2neg=2, sqrt=3, rsq=4, cos=5)
_QUERY_FN = ctypes.CFUNCTYPE(None, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float),
                             ctypes.POINTER(ctypes.c_float))


def install_hw_query(at):
    """Install the function that answers the gamma-Poisson class's questions (PRNB-7: nb_model.c, hwq_ask).
    `at(op, x)` returns the float32 array y[i] = HW_op(x[i]) computed ON THE DEVICE by the product's probe
    (prosstt_amd.device.Context.hw_math_at; op an integer of HW_OPS)."""
    global _HW_QUERY

    def answer(n, op_p, x_p, y_p):
        op = np.ctypeslib.as_array(op_p, shape=(n,))
        x = np.ctypeslib.as_array(x_p, shape=(n,))
        y = np.ctypeslib.as_array(y_p, shape=(n,))
        for o in np.unique(op):
            pick = np.flatnonzero(op == o)
            y[pick] = at(int(o), np.ascontiguousarray(x[pick]))

    _HW_QUERY = _QUERY_FN(answer)
    lib().prnb_set_hw_query(ctypes.cast(_HW_QUERY, ctypes.c_void_p))


def use_standin_query(on=True):
    """Test-only: answer the questions with the model's own libm stand-ins THROUGH the query machinery (the result must
    equal the direct evaluation bit for bit); False removes the query function."""
    global _HW_QUERY
    if on:
        lib().prnb_set_hw_query_standins()
    else:
        lib().prnb_set_hw_query(None)
    _HW_QUERY = None


def query_stats():
    """(rounds, values) of the last resolution of waiting samples."""
    r, v = ctypes.c_int64(0), ctypes.c_int64(0)
    lib().prnb_query_stats(ctypes.byref(r), ctypes.byref(v))
    return r.value, v.value


def install_hw_tables(probe):
    """Install the device's tables of the three hardware functions into the model (bit-exact mode).
    `probe(op, first_bits, count)` returns a float32 array y[i] = HW_op(as_float(first_bits + i)) computed ON THE
    DEVICE by the product's probe kernel (prosstt_amd.device.Context.hw_math; op in 'rcp', 'log2', 'exp2neg').
    Idempotent per process."""
    global _HW_TABLES
    if _HW_TABLES is not None:
        return
    tabs = {}
    for op, (first, count) in HW_RANGES.items():
        t = np.ascontiguousarray(probe(op, first, count), np.float32)
        assert t.shape == (count,)
        tabs[op] = t
    lib().prnb_set_hw_tables(tabs["rcp"].ctypes.data, tabs["log2"].ctypes.data, HW_RANGES["log2"][0], HW_RANGES["log2"][1],
                             tabs["exp2neg"].ctypes.data, HW_RANGES["exp2neg"][0], HW_RANGES["exp2neg"][1])
    _HW_TABLES = tabs


def install_hw_tables_from_device():
    """install_hw_tables with the product's probe on the current device (tests, smoke(), tools: GPU box only)."""
    if _HW_TABLES is not None:
        return
    from prosstt_amd import device
    ctx = device.get_context()
    install_hw_query(ctx.hw_math_at)
    install_hw_tables(ctx.hw_math)


def remove_hw_tables():
    global _HW_TABLES, _HW_QUERY
    lib().prnb_set_hw_tables(None, None, 0, 0, None, 0, 0)
    lib().prnb_set_hw_query(None)
    _HW_TABLES = None
    _HW_QUERY = None


def hw_mode():
    """True: the device's tables are installed (the model equals the kernels bit for bit);
    False: libm stand-ins (the same law to 1e-6; CPU-only law tests)."""
    return bool(lib().prnb_hw_mode())


def hw_math(op, x):
    """The model's view of a hardware function (through the installed table, or the libm stand-in)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().prnb_hw_math(dict(rcp=0, log2=1, exp2neg=2)[op], x, y, x.size)
    return y


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


def sample_selected(means, row_of_cell, scaling, alpha, beta, seed, cells, genes, cell_offset=0, cell_index=None):
    """(path, count) of the samples (cells[i], genes[i]): path 0 degenerate / 1 inversion / 2 gamma-Poisson."""
    means = np.ascontiguousarray(means, np.float32)
    rows, G = means.shape
    cells = np.ascontiguousarray(cells, np.int64)
    genes = np.ascontiguousarray(genes, np.int32)
    path, count = np.empty(cells.size, np.int32), np.empty(cells.size, np.int32)
    if cell_index is not None:
        cell_index = np.ascontiguousarray(cell_index, np.int64)
    lib().prnb_sample_selected(means, rows, G, np.ascontiguousarray(row_of_cell, np.int32),
                               np.ascontiguousarray(scaling, np.float64), np.ascontiguousarray(alpha, np.float64),
                               np.ascontiguousarray(beta, np.float64), seed, cell_offset,
                               cell_index.ctypes.data if cell_index is not None else None, cells, genes, cells.size,
                               path, count)
    return path, count


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
