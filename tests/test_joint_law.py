"""
Are the counts JOINTLY what the reference draws -- independent given the parameters
(/root/reference/prosstt/simulation.py:647-648: one ``rvs()`` over the flattened arrays)?

Every other law test of this suite is marginal (per-parameter chi-square, whole-matrix histogram and moments).  The
sampler keys Philox4x32-7 by (cell, gene quad) and hand-splits its first two rounds: a dependence between the four
words of a call, between neighbouring quads, gene tiles or cells would pass all of those.  tests/joint_law.py scores
products of standardised neighbours (~N(0, 1) each under independence) and the variance of row and column sums:

  * CPU (here, no GPU): the scalar model's matrix with the definition's 7 rounds AND with 10 rounds (Random123's
    default) -- the same scores, side by side; a generator change must keep them inside 5 sigma;
  * -m gpu: the WHOLE C3 matrix (1e9 counts) on the device, plus a second matrix whose cell ids are 2^32 higher
    (the counter's cell_hi word) correlated with the first sample by sample.
Mandatory for any change of generator, round count or counter layout.
"""
import numpy as np
import pytest
import torch

from joint_law import JointLaw, moments, report

SIGMA = 5.0


def _synthetic(seed, rows, G, N):
    rng = np.random.default_rng(seed)
    means = np.exp(rng.normal(np.log(1.2), 1.6, (rows, G))).astype(np.float32)     # medians about 1, a tail into the hundreds
    roc = rng.integers(0, rows, N).astype(np.int32)
    sc = np.exp(rng.normal(0, 0.7, N))
    al = np.exp(rng.normal(np.log(0.2), np.log(1.5), G))
    be = np.exp(rng.normal(0, np.log(1.5), G)) + 1
    return means, roc, sc, al, be


def _scores_of(X, means, roc, sc, al, be, X_pair=None, chunk=4096):
    dev = X.device
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=dev)
    M, R, S, A, B = t(means, torch.float32), t(roc, torch.int32), t(sc, torch.float64), t(al, torch.float64), t(be, torch.float64)
    jl = JointLaw(X.shape[1], dev)
    for lo in range(0, X.shape[0], chunk):
        hi = min(lo + chunk, X.shape[0])
        mu, var = moments(M, R, S, A, B, lo, hi)
        jl.add(X[lo:hi], mu, var, None if X_pair is None else X_pair[lo:hi])
    return jl.scores()


def test_model_matrix_is_jointly_independent_with_7_and_with_10_rounds():
    from oracle import nb_model
    means, roc, sc, al, be = _synthetic(11, 40, 2048, 3000)
    lines = []
    for rounds in (7, 10):
        with nb_model.philox_rounds_for_draws(rounds):
            X = nb_model.sample_counts(means, roc, sc, al, be, seed=20261003)
            X2 = nb_model.sample_counts(means, roc, sc, al, be, seed=20261003, cell_offset=1 << 32)
        scores = _scores_of(torch.as_tensor(X), means, roc, sc, al, be, X_pair=torch.as_tensor(X2))
        lines.append(report(scores, "C model, Philox4x32-%d, %d x %d" % (rounds, X.shape[0], X.shape[1])))
        for key, (tscore, detail) in scores.items():
            assert abs(tscore) < SIGMA, "%s with %d rounds: %+.2f sigma (%s)" % (key, rounds, tscore, detail)
        assert (X != X2).mean() > 0.2            # (the second matrix is another draw, not a copy)
    assert nb_model.count_rounds() == 7
    print("\n" + "\n".join(lines))


def test_the_statistic_sees_a_dependence():
    """The scores are not blind: a matrix whose odd genes repeat the uniform of their even neighbour (here: a copy of
    the count when the parameters are equal) is flagged by the inside-quad score, one whose rows are shifted copies by
    the cell score."""
    from oracle import nb_model
    rng = np.random.default_rng(3)
    G, N = 1024, 1500
    means = np.repeat(np.exp(rng.normal(np.log(2.0), 1.0, (8, G // 2))), 2, axis=1).astype(np.float32)
    roc = rng.integers(0, 8, N).astype(np.int32)
    sc, al, be = np.ones(N), np.full(G, 0.2), np.full(G, 2.0)
    X = nb_model.sample_counts(means, roc, sc, al, be, seed=5)
    bad = X.copy()
    bad[:, 1::2] = np.where(rng.random((N, G // 2)) < 0.05, X[:, 0::2], X[:, 1::2])      # 5 % of the odd genes copy their neighbour
    scores = _scores_of(torch.as_tensor(bad), means, roc, sc, al, be)
    assert scores["genes_lag1_inside_quad"][0] > 20 and abs(scores["cells_lag1"][0]) < SIGMA
    same_row = np.zeros(N, np.int32)
    Y = nb_model.sample_counts(means, same_row, sc, al, be, seed=6)
    Y[1::2] = np.where(rng.random((N // 2, G)) < 0.5, Y[0::2], Y[1::2])            # half of every odd cell copies the cell before
    scores = _scores_of(torch.as_tensor(Y), means, same_row, sc, al, be)
    assert scores["cells_lag1"][0] > 20 and scores["column_sums_variance"][0] > SIGMA


@pytest.mark.gpu
def test_whole_c3_matrix_is_jointly_independent():
    from prosstt_amd import device, workloads
    ctx = device.get_context()
    work = workloads.build("C3")
    N = work.cfg["N"]
    pt, br, sc, rows = work.plan(N)
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=20261003)
    X2 = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=20261003, cell_offset=1 << 32)
    scores = _scores_of(X, means, rows, sc, work.alpha, work.beta, X_pair=X2, chunk=2048)
    print("\n" + report(scores, "device, whole C3 matrix, %d x %d (+ the matrix of cell ids 2^32 higher)" % (N, work.tree.G)))
    for key, (tscore, detail) in scores.items():
        assert abs(tscore) < SIGMA, "%s: %+.2f sigma (%s)" % (key, tscore, detail)
    # the second matrix is another draw of the same law
    assert 0.2 < float((X[:2048] != X2[:2048]).float().mean()) < 0.6


@pytest.mark.gpu
def test_32_branch_matrix_is_jointly_independent():
    """The same on north_star's shape (32-branch tree, 50 000 x 20 000)."""
    from prosstt_amd import device, workloads
    ctx = device.get_context()
    work = workloads.build("T32")
    pt, br, sc, rows = work.plan()
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=77)
    scores = _scores_of(X, means, rows, sc, work.alpha, work.beta, chunk=2048)
    print("\n" + report(scores, "device, T32 matrix, %d x %d" % X.shape))
    for key, (tscore, detail) in scores.items():
        assert abs(tscore) < SIGMA, "%s: %+.2f sigma (%s)" % (key, tscore, detail)


@pytest.mark.gpu
def test_c4_matrix_is_jointly_independent():
    """And at 4e9 counts (C4: the 32-branch tree, 200 000 x 20 000), the cells presented grouped by mean-tensor row as
    the drop-in API presents them -- neighbouring rows of the device matrix then share their parameters, which is where
    a dependence between neighbouring cell counters would show most."""
    from prosstt_amd import device, workloads
    ctx = device.get_context()
    work = workloads.build("C4")
    pt, br, sc, rows = work.plan()
    means = work.tree.device_means()
    order = device.plan_order(rows, means.shape[0])
    rows, sc = rows[order], sc[order]
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=4004, cell_index=order.astype(np.int64))
    scores = _scores_of(X, means, rows, sc, work.alpha, work.beta, chunk=2048)
    print("\n" + report(scores, "device, C4 matrix (cells grouped by row), %d x %d" % X.shape))
    for key, (tscore, detail) in scores.items():
        assert abs(tscore) < SIGMA, "%s: %+.2f sigma (%s)" % (key, tscore, detail)
