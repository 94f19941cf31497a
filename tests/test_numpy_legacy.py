"""
Pins oracle/numpy_legacy.c -- the C restatement of numpy's legacy MT19937 ->
gamma -> Poisson negative-binomial chain behind the reference's
scipy.stats.nbinom(n=r, p=1-p).rvs() (simulation.py:647-648) -- bit-identical to numpy
itself, and to the oracle's numpy path on a real draw_counts call.
"""
import numpy as np

from conftest import load_golden
from oracle.numpy_legacy import LegacyState


def test_streams_bit_identical_to_numpy():
    for seed in (0, 92, 123456, 2 ** 32 - 1):
        np.random.seed(seed)
        s = LegacyState(seed)
        np.testing.assert_array_equal(np.random.random_sample(1000), s.random_sample(1000))
        np.testing.assert_array_equal(np.random.standard_normal(1001), s.standard_normal(1001))
        for shape in (0.05, 0.5, 1.0, 1.7, 30.0):
            np.testing.assert_array_equal(np.random.standard_gamma(shape, 500), s.standard_gamma(shape, 500))
        lam = np.exp(np.random.uniform(-3, 9, 5000))
        s.random_sample(5000)
        np.testing.assert_array_equal(np.random.poisson(lam), s.poisson(lam))
        n = np.exp(np.random.uniform(-4, 4, 20000))
        p = np.random.uniform(0.001, 0.999, 20000)
        s.random_sample(40000)
        np.testing.assert_array_equal(np.random.negative_binomial(n, p), s.negative_binomial(n, p))


def test_draw_counts_bit_identical_to_reference_fixture():
    """The whole of draw_counts in C from the reference's seed reproduces the reference's X."""
    g = load_golden("g6_sampling_bifurcation")
    order = ["A", "B", "C"]
    means = np.concatenate([g["means_%s" % b] for b in order])
    starts = {"A": 0, "B": 40, "C": 40}
    rows = np.array([40 * order.index(b) + p - starts[b] for p, b in zip(g["pt"], g["br"])], np.int64)
    N = len(rows)
    seed = 360 + 1                      # make_golden.py: np.random.seed(seed + 1) before sample_density
    s = LegacyState(seed)
    s.random_sample(N)                  # np.random.choice consumed N uniforms ...
    sc = np.exp(s.standard_normal(N) * 0.7 + 0.0)   # ... calc_scalings N normals
    np.testing.assert_array_equal(sc, g["scalings"])
    X = s.draw_counts(means, rows, sc, g["alpha"], g["beta"])
    np.testing.assert_array_equal(X, g["X"])
