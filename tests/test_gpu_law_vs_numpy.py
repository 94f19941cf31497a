"""
-m gpu: the device counts against the reference's OWN sampler on a whole workload.

For the (mean, alpha, beta) of every (cell, gene) of a C2-sized plan (3-branch tree through the
product pipeline, 2000 cells x 5000 genes = 1e7 samples) draw one matrix on the device and one with
numpy's RandomState.negative_binomial -- what scipy.stats.nbinom(n=r, p=1-p).rvs() of
simulation.py:647-648 calls.  The two streams differ, the laws must not:
  * pooled count histogram: two-sample chi-square;
  * per-gene totals: z-scores of the difference ~ N(0,1) (mean, sd, extreme);
  * per-cell totals likewise.
"""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu


def test_device_matrix_vs_numpy_negative_binomial():
    from prosstt_amd import device, workloads
    from oracle import ref_numpy
    ctx = device.get_context()
    work = workloads.build("C2")
    pt, br, sc, rows = work.plan(2000)
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=31337).cpu().numpy().astype(np.int64)
    mu = means.cpu().numpy().astype(np.float64)[rows] * sc[:, None]
    p, r = ref_numpy.get_pr_umi(work.alpha[None, :], work.beta[None, :], mu)
    np.random.seed(4)
    Y = np.random.negative_binomial(r, 1 - p)                      # the reference's sampler
    assert X.shape == Y.shape == (2000, 5000)
    # pooled histogram
    kmax = 400
    hx = np.bincount(np.minimum(X, kmax).ravel(), minlength=kmax + 1).astype(float)
    hy = np.bincount(np.minimum(Y, kmax).ravel(), minlength=kmax + 1).astype(float)
    keep = (hx + hy) >= 40
    hx2 = np.append(hx[keep], hx[~keep].sum())
    hy2 = np.append(hy[keep], hy[~keep].sum())
    chi2, pval, _, _ = stats.chi2_contingency(np.vstack([hx2, hy2]))
    assert pval > 1e-4, (chi2, pval)
    assert abs((X == 0).mean() - (Y == 0).mean()) < 5e-4
    # per-gene and per-cell totals: difference of two independent draws of the same law
    var = work.alpha[None, :] * mu ** 2 + work.beta[None, :] * mu
    for axis in (0, 1):
        z = (X.sum(axis=axis) - Y.sum(axis=axis)) / np.sqrt(2 * var.sum(axis=axis))
        assert abs(z.mean()) < 5 / np.sqrt(len(z)), (axis, z.mean())
        assert abs(z.std() - 1) < 0.08, (axis, z.std())
        assert np.abs(z).max() < 6
    # both agree with the analytic first two moments
    assert abs(X.sum() / mu.sum() - 1) < 2e-3 and abs(Y.sum() / mu.sum() - 1) < 2e-3
    sel = mu >= 0.05      # standardised second moment, 1 in expectation (tiny means make it too heavy-tailed)
    rx = (((X - mu) ** 2) / var)[sel].mean()
    ry = (((Y - mu) ** 2) / var)[sel].mean()
    assert abs(rx - 1) < 0.03 and abs(ry - 1) < 0.03 and abs(rx - ry) < 0.03, (rx, ry)
