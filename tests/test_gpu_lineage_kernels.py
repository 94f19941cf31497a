"""
-m gpu: the lineage kernels (K2) against numpy float64 restatements of
simulation.py:269-272 / sim_utils.py:145-168 / tree.py:181-182, through the C ABI.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from prosstt_amd import device
    return device.get_context()


def walk(rng, T, K):
    return rng.normal(0, 0.1, (T, K)).cumsum(axis=0) + np.log(rng.uniform(0, 1.5, K))


@pytest.mark.parametrize("T,K,G,sib_T", [(50, 25, 20000, [50]), (40, 5, 64, []), (33, 7, 1001, [25, 60, 33]), (50, 40, 3000, [50, 20]), (3, 70, 130, [1]),
                                         (2, 2, 5, [2])])
def test_attempt_matches_numpy(ctx, T, K, G, sib_T):
    import torch
    from oracle import ref_numpy
    rng = np.random.default_rng(T * 1000 + K)
    P = walk(rng, T, K)
    H = rng.standard_gamma(0.05, (K, G))
    H[:, G // 2] = 0.0                       # a constant gene: Pearson r is NaN, never < 0
    sibs = [walk(rng, t, K) for t in sib_T]
    Hd = torch.as_tensor(H, device=ctx.torch_device)
    mx, counts = ctx.lineage_attempt(P, Hd, sibs)
    rel = P @ H
    assert mx == pytest.approx(rel.max(), rel=1e-13, abs=1e-13)
    assert len(counts) == len(sibs)
    for j, S in enumerate(sibs):
        c = min(T, S.shape[0])
        r = ref_numpy.pearson_columns(rel[:c], (S @ H)[:c])
        with np.errstate(invalid="ignore"):
            want = int(np.sum(r < 0))
            ties = int(np.sum(np.abs(r) < 1e-12))
        assert abs(counts[j] - want) <= ties


def test_commit_and_means(ctx):
    import torch
    rng = np.random.default_rng(3)
    T, K, G = 50, 25, 5003
    H = rng.standard_gamma(0.05, (K, G))
    Hd = torch.as_tensor(H, device=ctx.torch_device)
    rel_d = torch.empty((2 * T, G), dtype=torch.float64, device=ctx.torch_device)
    gmax = torch.full((G,), -np.inf, dtype=torch.float64, device=ctx.torch_device)
    P1, P2 = walk(rng, T, K), walk(rng, T, K)
    ctx.lineage_commit(P1, Hd, rel_d[:T], gmax)
    ctx.lineage_commit(P2, Hd, rel_d[T:], gmax)
    rel = np.concatenate([P1 @ H, P2 @ H])
    np.testing.assert_allclose(rel_d.cpu().numpy(), rel, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(gmax.cpu().numpy(), rel.max(axis=0), rtol=1e-13, atol=1e-13)
    base = np.exp(rng.normal(0.8, 1, G))
    means = ctx.means_from_rel(rel_d, torch.as_tensor(base, device=ctx.torch_device)).cpu().numpy()
    want = np.exp(rel) * base                      # tree.py:181-182 in float64
    assert means.dtype == np.float32
    np.testing.assert_allclose(means, want, rtol=1.2e-7, atol=1.2e-38)   # one binary32 rounding


def test_device_mode_walk_bit_exact_and_usable(ctx):
    """K1 against the C model (bit-exact), and simulate_lineage(rng='device') end to end."""
    from oracle import nb_model
    for T, K, seed, sid in ((50, 25, 7, 0), (2, 3, 8, (5 << 32) | 9), (41, 70, 2 ** 63 + 5, 1)):
        got = ctx.lineage_walk(seed, sid, T, K)
        np.testing.assert_array_equal(got, nb_model.lineage_walk(seed, sid, T, K))
    from prosstt_amd import simulation as sim
    from prosstt_amd import tree as ptree
    np.random.seed(3)
    t = ptree.Tree(G=300, modules=6)
    stats = []
    rel, prog, H = sim.simulate_lineage(t, a=0.05, rng="device", seed=11, stats=stats)
    assert [str(b) for b in rel.index] == ["A", "B", "C"] and rel["B"].shape == (40, 300)
    # the accepted programs are the model's walks for (branch ordinal, attempt), shifted onto the parent
    tries = {}
    for b, _, _ in stats:
        tries[str(b)] = tries.get(str(b), 0) + 1
    root = nb_model.lineage_walk(11, (0 << 32) | (tries["A"] - 1), 40, 6)
    np.testing.assert_array_equal(prog["A"], root)
    child = nb_model.lineage_walk(11, (1 << 32) | (tries["B"] - 1), 40, 6)
    np.testing.assert_array_equal(prog["B"], child - (child[0] - root[-1]))
    np.testing.assert_allclose(rel["C"], prog["C"] @ H, rtol=1e-12, atol=1e-13)
    np.random.seed(3)                                                # same coefficients -> same accepted attempts
    again = sim.simulate_lineage(t, a=0.05, rng="device", seed=11)
    np.testing.assert_array_equal(again[1]["A"], prog["A"])


def test_means_underflow_stays_positive(ctx):
    """exp(rel)*base far below binary32's range: the reference's float64 mean is positive, so the
    stored mean must be too (an exact 0 would raise the reference's ValueError in strict mode)."""
    import torch
    rel = torch.tensor([[-200.0, -80.0, 0.0, -800.0]], dtype=torch.float64, device=ctx.torch_device)
    base = torch.tensor([1.0, 1.0, 2.0, 1.0], dtype=torch.float64, device=ctx.torch_device)
    m = ctx.means_from_rel(rel, base).cpu().numpy()[0]
    assert m[0] == np.float32(1.17549435e-38) and m[1] == np.float32(np.exp(-80.0)) and m[2] == 2.0
    assert m[3] == 0.0            # exp(-800) is 0 in binary64 as well: a true zero stays zero
    X = ctx.sample_counts(m[None, :3].copy(), np.zeros(5, np.int32), np.ones(5), np.full(3, 0.2), np.full(3, 2.0), seed=1)
    assert X.shape == (5, 3)      # strict mode accepts it; the tiny means just give zeros
    assert int(X[:, :2].sum()) == 0
