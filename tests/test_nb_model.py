"""
CPU tests of oracle/nb_model.c, the scalar C model of the device count sampler (PRNB-7; here on its libm stand-ins for the hardware functions unless a GPU is present):
known-answer vectors of Philox4x32-10 and -7, accuracy of the deterministic binary32 math, and the
LAW of the sampler against the reference's distribution (scipy.stats.nbinom tables of fixture
g7 and the oracle's numpy path).  The HIP kernel is then held bit-exact to this model (-m gpu).
"""
import numpy as np
import pytest
from scipy import stats

from conftest import load_golden
from oracle import nb_model as nm
from oracle import ref_numpy


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert list(nm.philox(ctr, key)) == want
        assert list(nm.philox_rounds(10, ctr, key)) == want


def test_philox_seven_rounds_is_the_count_samplers_generator():
    """The count sampler draws from Philox4x32-7 (Salmon et al. 2011: the fewest rounds that pass
    BigCrush).  First vector: Random123's kat_vectors for philox4x32 7; the others: the same round
    function, whose 10-round form reproduces all three 10-round vectors above.  Then plain sanity on
    a million words of the 7-round stream (bit balance, byte chi-square, lag-1 serial correlation of
    neighbouring counters) -- not a substitute for BigCrush, a guard against a broken build."""
    assert nm.count_rounds() == 7
    kat7 = [([0, 0, 0, 0], [0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
            ([0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
            ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
             [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a])]
    for ctr, key, want in kat7:
        assert list(nm.philox_rounds(7, ctr, key)) == want
    # the sampler's own use of it: counter (cell, 0, gene >> 2, 0), one word per gene
    words = np.concatenate([nm.philox_rounds(7, [c, 0, g, 0], [12345, 678]) for c in range(500) for g in range(500)])
    n = words.size
    bits = np.unpackbits(words.view(np.uint8))
    assert abs(bits.mean() - 0.5) < 4 * 0.5 / np.sqrt(bits.size)
    counts = np.bincount(words.view(np.uint8), minlength=256)
    chi = ((counts - n * 4 / 256) ** 2 / (n * 4 / 256)).sum()
    assert chi < 255 + 5 * np.sqrt(2 * 255)
    u = words.astype(np.float64) / 2 ** 32
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5 / np.sqrt(n)
    assert abs(np.corrcoef(u[:-4:4], u[4::4])[0, 1]) < 5 / np.sqrt(n / 4)      # same word of neighbouring gene quads


def test_deterministic_math_accuracy():
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(-40, 40, 400000)).astype(np.float32)
    assert np.max(np.abs(nm.math("rcp", x) * x.astype(np.float64) - 1)) < 1.2e-7
    ref = np.log(x.astype(np.float64))
    assert np.max(np.abs(nm.math("log", x) - ref) / np.maximum(np.abs(ref), 1e-3)) < 2e-7
    t = np.concatenate([x[x < 1e6], rng.uniform(0, 2, 200000).astype(np.float32)])
    assert np.max(np.abs(nm.math("log1p", t) / np.log1p(t.astype(np.float64)) - 1)) < 2.5e-7
    e = -rng.uniform(0, 87, 400000).astype(np.float32)
    assert np.max(np.abs(nm.math("exp", e) / np.exp(e.astype(np.float64)) - 1)) < 2e-7
    assert nm.math("exp", np.array([-88.0, -1e9], np.float32)).tolist() == [0.0, 0.0]
    w = rng.integers(0, 2 ** 32, 400000, dtype=np.uint64).astype(np.uint32)
    c = nm.math("cos2pi", w.view(np.float32))
    assert np.max(np.abs(c - np.cos(2 * np.pi * w.astype(np.float64) / 2 ** 32))) < 3e-7
    d = rng.uniform(-0.9, 3, 400000).astype(np.float32)
    ref = np.log1p(d.astype(np.float64)) - d.astype(np.float64)
    assert np.max(np.abs(nm.math("log1pmx", d) - ref) / np.abs(ref).clip(1e-30)) < 3e-6
    u = nm.math("unif", np.array([0, 1, 2 ** 31, 2 ** 32 - 1], np.uint32).view(np.float32))
    assert u[0] > 0 and u[-1] <= 1.0 and abs(u[2] - 0.5) < 1e-7


def pooled_chi2(x, pmf, n):
    kmax = len(pmf)
    cnt = np.bincount(np.minimum(x, kmax - 1), minlength=kmax).astype(float)
    exp = pmf * n
    exp[-1] += n * max(0.0, 1 - pmf.sum())
    obs_b, exp_b, co, ce = [], [], 0.0, 0.0
    for k in range(kmax):
        co += cnt[k]
        ce += exp[k]
        if ce >= 50:
            obs_b.append(co); exp_b.append(ce); co = ce = 0.0
    obs_b[-1] += co
    exp_b[-1] += ce
    obs_b, exp_b = np.array(obs_b), np.array(exp_b)
    return stats.chi2.sf(((obs_b - exp_b) ** 2 / exp_b).sum(), len(obs_b) - 1)


def test_law_against_scipy_tables():
    """Each (m, a, b) of fixture g7 (light path, both Poisson branches of the heavy path, the
    Poisson limit beta = 1+1e-8, r < 1, alpha = 0): chi-square vs nbinom(n=r, p=1-p).pmf,
    mean within 6 SE, variance within 8 %."""
    g = load_golden("g7_nb_tables")
    n = 1_000_000
    pvals = []
    for i, (m, a, b, r, p, mean, var) in enumerate(g["params"]):
        x = nm.sample_iid(m, a, b, n, seed=1234 + i, gene=i)
        assert abs(x.mean() - mean) < 6 * np.sqrt(var / n) + 2e-6 * mean, (m, a, b)
        # (the sample variance of a million draws of a tiny mean rests on n * m nonzero counts: 6 of its own standard errors)
        assert abs(x.var() / var - 1) < max(0.08, 6.0 / np.sqrt(n * min(mean, 1.0))), (m, a, b)
        pvals.append(pooled_chi2(x, g["pmf"][i].copy(), n))
    assert min(pvals) > 1e-4 / len(pvals), pvals


def test_the_query_machinery_reproduces_the_direct_evaluation():
    """PRNB-7's gamma-Poisson class takes its transcendentals from the device: the model ASKS for them, level by level,
    and replays each waiting sample's tape (nb_model.c: hwq_ask / resolve_waiting).  With the model's own libm stand-ins
    answering THROUGH that machinery, every count must equal the direct evaluation, bit for bit -- on a matrix with a
    third of its samples in that class (both Poisson branches, the boost below r = 1, rejected attempts) -- and the
    questions come in rounds of dependent levels, not one by one."""
    rng = np.random.default_rng(1)
    G = 64
    means = np.exp(rng.normal(3.0, 2.0, (5, G))).astype(np.float32)
    roc = rng.integers(0, 5, 300).astype(np.int32)
    sc = np.exp(rng.normal(0, 0.7, 300))
    al = np.exp(rng.normal(np.log(0.2), 1.0, G))
    be = 1 + np.exp(rng.normal(0, 1.5, G))
    al[:4] = [0, 1e-9, 3.0, 0.0]
    be[:4] = [1 + 1e-8, 1 + 1e-8, 40.0, 1.00001]
    assert not nm.hw_mode() or pytest.skip("the device's tables are installed in this process")
    direct = nm.sample_counts(means, roc, sc, al, be, 7)
    path = nm.nb_params(means, roc, sc, al, be)[3]
    heavy = int((path == 2).sum())
    assert heavy > 0.2 * path.size
    try:
        nm.use_standin_query(True)
        asked = nm.sample_counts(means, roc, sc, al, be, 7)
        rounds, values = nm.query_stats()
        iid = nm.sample_iid(200.0, 0.3, 2.0, 50000, seed=3)
    finally:
        nm.use_standin_query(False)
    np.testing.assert_array_equal(asked, direct)
    assert 4 <= rounds <= 80 and 4 * heavy < values < 40 * heavy
    np.testing.assert_array_equal(iid, nm.sample_iid(200.0, 0.3, 2.0, 50000, seed=3))


def test_zero_fraction_matches_closed_form():
    """P(X = 0) = (1+theta)^(-r) over a grid of small means (the dominant outcome: ~50 % zeros)."""
    n = 400_000
    for m, a, b in [(0.01, 0.2, 2.0), (0.3, 0.1, 1.5), (2.0, 0.5, 3.0), (11.0, 0.2, 2.0), (25.0, 0.3, 2.0)]:
        theta = a * m + b - 1
        p0 = (1 + theta) ** (-m / theta)
        x = nm.sample_iid(m, a, b, n, seed=5)
        assert abs((x == 0).mean() - p0) < 5 * np.sqrt(p0 * (1 - p0) / n) + 1e-6


def test_matrix_entry_is_pure_function_of_global_cell_and_gene():
    rng = np.random.default_rng(3)
    means = np.exp(rng.normal(0.5, 1.5, (9, 40))).astype(np.float32)
    roc = rng.integers(0, 9, 50).astype(np.int32)
    sc = np.exp(rng.normal(0, 0.7, 50))
    al, be = np.full(40, 0.2), np.full(40, 2.0)
    full = nm.sample_counts(means, roc, sc, al, be, 11)
    part = nm.sample_counts(means, roc[20:], sc[20:], al, be, 11, cell_offset=20)
    np.testing.assert_array_equal(full[20:], part)
    idx = np.array([7, 3, 44])
    picked = nm.sample_counts(means, roc[idx], sc[idx], al, be, 11, cell_index=idx)
    np.testing.assert_array_equal(full[idx], picked)
    assert not np.array_equal(full, nm.sample_counts(means, roc, sc, al, be, 12))


def test_model_vs_reference_draw_on_a_real_tree():
    """Same (means, plan) as fixture g6: the model's matrix and the reference's matrix (numpy
    stream) agree in total, zero fraction and per-gene means within sampling error; the
    deterministic (mu, p, r) agree to binary32 rounding."""
    g = load_golden("g6_sampling_unequal")
    order = ["A", "B", "C", "D", "E"]
    means = np.concatenate([g["means_%s" % b] for b in order])
    offsets = np.cumsum([0] + [g["means_%s" % b].shape[0] for b in order])
    starts = {"A": 0, "B": 70, "C": 70, "D": 130, "E": 130}
    rows = np.array([offsets[order.index(b)] + p - starts[b] for p, b in zip(g["pt"], g["br"])], np.int32)
    mu, p, r, path = nm.nb_params(means.astype(np.float32), rows, g["scalings"], g["alpha"], g["beta"])
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-6)
    np.testing.assert_allclose(p, g["p"], rtol=1e-6)
    np.testing.assert_allclose(r, g["r"], rtol=2e-6)
    reps = 200
    tot = np.array([nm.sample_counts(means.astype(np.float32), rows, g["scalings"], g["alpha"], g["beta"], s).sum()
                    for s in range(reps)])
    var = (g["alpha"] * g["mu"] ** 2 + g["beta"] * g["mu"]).sum()
    assert abs(tot.mean() - g["mu"].sum()) < 5 * np.sqrt(var / reps)
    assert abs(g["X"].sum() - g["mu"].sum()) < 5 * np.sqrt(var)
    assert abs(tot.std() / np.sqrt(var) - 1) < 0.25


def test_degenerate_parameters():
    z = nm.sample_iid(0.0, 0.2, 2.0, 100)
    assert not z.any()
    assert not nm.sample_iid(3.0, 0.0, 1.0, 100).any()          # alpha=0, beta=1: zeros, as in the reference
    assert not nm.sample_iid(3.0, 0.0, 0.5, 100).any()          # theta < 0: 0 (the wrapper raises ValueError)
    big = nm.sample_iid(3.0e5, 0.3, 2.0, 2000, seed=1)           # far beyond abs_max=5000 x scaling
    assert abs(big.mean() / 3.0e5 - 1) < 0.05 and big.min() >= 0


def test_inversion_class_rule():
    """PRNB-7 (as PRNB-6): inversion iff theta = a*m + b - 1 <= 24 and t2 = -log2 P(X=0) = m*log2(1+theta)/theta < 27.4112 (t < 19)
    (P0 * 2^32 >= 24, tail ratio <= 24/25); m <= 0 or theta <= 0 is the degenerate path.  Both classes
    follow the same law, so the split must not show in the moments."""
    means = np.array([[0.5, 18.9, 30.0, 60.0, 6.0, 8.0, 12.5, 3.0, 3.0, 3.0, 25.0, 2.0, 2.0, 0.0, 75.0, 80.0, 25.0, 19.5]], np.float32)
    alpha = np.array([0.2, 0.2, 0.2, 0.2, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0, -0.5, 0.3, np.nan, 0.2, 0.1, 0.1, 0.0, 0.0])
    beta = np.array([2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 17.5, 25.0, 25.5, 38.0, 1.0, 2.0, 2.0, 3.0, 3.0, 1.0 + 1e-8, 1.0 + 1e-8])
    path = nm.nb_params(means, np.zeros(1, np.int32), np.ones(1), alpha, beta)[3][0]
    #         m=.5 18.9 30 60 (t=.4, 8.9, 8.9, 12.2) | a=2: theta=13, 17, 26 | b-1 = 16.5, 24, 24.5 | a<0: theta=24.5 | b=1: theta=.6
    #         | NaN | m=0 | t=18.6, 19.2 | Poisson limit: t = m = 25 / 19.5
    assert path.tolist() == [1, 1, 1, 1, 1, 1, 2, 1, 1, 2, 2, 1, 0, 0, 1, 2, 2, 2]
    for m, a, b in ((7.4, 2.0, 2.0), (7.6, 2.0, 2.0), (11.4, 2.0, 2.0), (11.6, 2.0, 2.0), (75.0, 0.1, 3.0), (80.0, 0.1, 3.0),
                    (60.0, 0.2, 2.0), (135.0, 0.163, 2.0), (140.0, 0.157, 2.0)):
        x = nm.sample_iid(m, a, b, 400000, seed=5)
        var = a * m * m + b * m
        assert abs(x.mean() - m) < 5 * np.sqrt(var / x.size)
        assert abs(x.var() / var - 1) < 0.03


def test_lost_mass_goes_to_the_tail():
    """A walk that runs out of representable pmf before the uniform is used up ends at the tail value
    where it stopped, never at 0: no zero may appear among samples whose P(X=0) is below 1e-7."""
    x = nm.sample_iid(75.0, 0.1, 3.0, 2_000_000, seed=9)        # P0 = e^-18.6
    assert x.min() > 0
    x = nm.sample_iid(25.0, 0.02, 1.3, 2_000_000, seed=9)       # P0 = e^-18.4, near-Poisson
    assert x.min() > 0


def test_device_mode_walk_law():
    """PRLW-1 (device-mode expression programs) follows simulation.diffusion's law
    (simulation.py:104-121): start = log(1.5 U), vel0 ~ N(0, 0.2^2), eta ~ U(0,1),
    increments eta*vel + N(0, (2/T)^2); and it is a pure function of (seed, stream, k)."""
    T, K = 50, 4000
    w = nm.lineage_walk(123, (3 << 32) | 7, T, K)
    assert w.shape == (T, K) and np.isfinite(w).all()
    start = w[0]
    u = np.exp(start) / 1.5
    assert stats.kstest(u, "uniform").pvalue > 1e-4
    vel0 = w[1] - w[0]
    assert abs(vel0.mean()) < 5 * 0.2 / np.sqrt(K) and abs(vel0.std() / 0.2 - 1) < 0.05
    assert stats.kstest(vel0 / 0.2, "norm").pvalue > 1e-4
    v = np.diff(w, axis=0)                                  # velocities vel[0..T-2]
    # regress vel[t+1] on vel[t] per walk: slope = eta in (0,1), residual sd = 2/T
    eta = (v[1:] * v[:-1]).sum(axis=0) / (v[:-1] ** 2).sum(axis=0)
    assert 0.35 < np.median(eta) < 0.65
    resid = v[1:] - eta * v[:-1]
    assert abs(resid.std() / (2 / T) - 1) < 0.05
    np.testing.assert_array_equal(w[:, :7], nm.lineage_walk(123, (3 << 32) | 7, T, 7))     # independent of K
    assert not np.array_equal(w[:, :7], nm.lineage_walk(123, (3 << 32) | 8, T, 7))
    assert not np.array_equal(w[:, :7], nm.lineage_walk(124, (3 << 32) | 7, T, 7))
