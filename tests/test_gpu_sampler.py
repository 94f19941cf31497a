"""
-m gpu: the HIP count sampler (K3) against the oracle, through the C ABI.

 * counts: BIT-EXACT against oracle/nb_model.c (the scalar C model of PRNB-7, reading the device's own tables of v_rcp/v_log/v_exp and asking it for the gamma-Poisson class's transcendentals) on the
   same seeded inputs -- integer work, no tolerance;
 * (mu, p, r): bit-exact against the model AND within rtol 1e-6 of the reference's
   float64 get_pr_umi (count_model.py:156-158) -- the stated fp32 tolerance;
 * the law: chi-square of device draws against scipy's NB pmf tables (fixture g7);
 * size-independent properties at larger sizes: chunking over cells and
   cell_offset invariance, run-to-run determinism, sum(X)/sum(mu) -> 1.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from prosstt_amd import device
    return device.get_context()


def synthetic(seed, rows, G, N, heavy_frac=None):
    """Inputs with the value mix SURVEY section 6 measured (median mu ~1.5, 90 % < 10, max ~1e4)."""
    rng = np.random.default_rng(seed)
    base = np.exp(rng.normal(0.8, 1.0, G))
    rel = rng.normal(0.0, 0.6, (rows, G)).cumsum(axis=0) * 0.15
    means = (np.exp(rel) * base).astype(np.float32)
    if heavy_frac:
        means[:, rng.random(G) < heavy_frac] *= 200.0
    row_of_cell = rng.integers(0, rows, N).astype(np.int32)
    scaling = np.exp(rng.normal(0, 0.7, N))
    alpha = np.exp(rng.normal(np.log(0.2), np.log(1.5), G))
    beta = np.exp(rng.normal(np.log(1.0), np.log(1.5), G)) + 1
    return means, row_of_cell, scaling, alpha, beta


@pytest.mark.parametrize("rows,G,N,seed", [
    (30, 256, 64, 1),        # exactly one gene tile
    (150, 1000, 333, 2),     # ragged in both directions, G % 256 != 0
    (7, 37, 5, 3),           # G % 4 != 0 -> scalar load/store path
    (40, 4, 1, 4),           # one cell, one quad
    (400, 2048, 777, 5),
])
def test_counts_bit_exact_vs_model(ctx, rows, G, N, seed):
    from oracle import nb_model
    means, roc, sc, al, be = synthetic(seed, rows, G, N, heavy_frac=0.05)
    got = ctx.sample_counts(means, roc, sc, al, be, seed=seed * 7919, cell_offset=seed * 10**6).cpu().numpy()
    want = nb_model.sample_counts(means, roc, sc, al, be, seed * 7919, seed * 10**6)
    assert got.dtype == np.int32 and got.shape == (N, G)
    np.testing.assert_array_equal(got, want)


def test_params_vs_model_and_reference(ctx):
    from oracle import nb_model, ref_numpy
    means, roc, sc, al, be = synthetic(11, 60, 500, 200, heavy_frac=0.1)
    mu, p, r, path = [t.cpu().numpy() for t in ctx.nb_params(means, roc, sc, al, be)]
    mmu, mp, mr, mpath = nb_model.nb_params(means, roc, sc, al, be)
    np.testing.assert_array_equal(mu, mmu)
    np.testing.assert_array_equal(p, mp)
    np.testing.assert_array_equal(r, mr)
    np.testing.assert_array_equal(path, mpath)
    assert set(np.unique(path)) == {1, 2}
    # float64 reference formulas on the same (fp32-stored) means
    mu64 = means.astype(np.float64)[roc] * sc[:, None]
    p64, r64 = ref_numpy.get_pr_umi(al[None, :], be[None, :], mu64)
    np.testing.assert_allclose(mu, mu64, rtol=1e-6)
    np.testing.assert_allclose(p, p64, rtol=1e-6)
    np.testing.assert_allclose(r, r64, rtol=2e-6)


def test_edge_parameters_bit_exact(ctx):
    """Poisson limit (beta = 1 + 1e-8, examples/linear.ipynb), alpha = 1e-4
    (many_branches_cells.ipynb cell 11), huge alpha, tiny and large means."""
    from oracle import nb_model
    G = 64
    means = np.tile(np.array([1e-7, 1e-3, 0.3, 2.0, 11.99, 12.01, 40.0, 3000.0], np.float32), (3, G // 8))
    means[1] *= 0.5
    means[2] *= 7.0
    roc = np.arange(300, dtype=np.int32) % 3
    sc = np.exp(np.random.default_rng(5).normal(0, 0.7, 300))
    for al, be in ((0.0, 1 + 1e-8), (1e-4, 1.3), (5.0, 2.0), (0.2, 40.0), (0.0, 7.0)):
        a, b = np.full(G, al), np.full(G, be)
        got = ctx.sample_counts(means, roc, sc, a, b, seed=99).cpu().numpy()
        want = nb_model.sample_counts(means, roc, sc, a, b, 99)
        np.testing.assert_array_equal(got, want)
        assert got.max() > 0


def test_inversion_class_limit_and_invalid_inputs_bit_exact(ctx, monkeypatch):
    """The inversion class is `theta <= 24 and -log P0 <= 19`; everything else -- including
    non-positive, infinite and NaN means, alpha < 0, beta < 1, beta - 1 > 24 -- takes the
    gamma-Poisson kernel or is a 0 by definition (genes whose theta can pass 24 at small means
    skip the zero test: prnb::zero_test_factor).  Unchecked mode, bit-exact
    against the model; the path codes of nb_params agree with the model's too."""
    from oracle import nb_model
    rng = np.random.default_rng(77)
    alphas = np.array([0.0, 1e-4, 0.2, 0.79, 0.8, 1.0, 3.0, 50.0, -0.1, -2.0, np.nan, 0.3])
    betas = np.array([1.0, 1 + 1e-8, 0.5, 2.0, 3.0, 16.9, 17.0, 17.1, 24.9, 25.0, 25.1, 40.0, 1.5, np.nan])
    G = len(alphas) * len(betas)                       # 168: ragged against the 256-gene tile, G % 4 == 0
    al, be = [x.ravel() for x in np.meshgrid(alphas, betas, indexing="ij")]
    rows = 24
    means = np.exp(rng.normal(0.5, 2.0, (rows, G))).astype(np.float32)
    means[0, :] = 0.0
    means[1, ::3] = -1.0
    means[2, ::5] = np.inf
    means[3, ::7] = np.nan
    means[4, :] = 1e-45                                # binary32 denormal
    means[5, :] = 18.999
    means[6, :] = 19.0
    means[7, :] = 19.001
    N = 600
    roc = rng.integers(0, rows, N).astype(np.int32)
    sc = np.exp(rng.normal(0, 0.5, N))
    sc[::50] = 1.0                                     # rows 5-7 hit the limit exactly for some cells
    want = nb_model.sample_counts(means, roc, sc, al, be, 4242, 17)
    got = ctx.sample_counts(means, roc, sc, al, be, seed=4242, cell_offset=17, check_domain=False).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert want.max() > 0 and (want[roc == 0] == 0).all()
    path = ctx.nb_params(means, roc, sc, al, be)[3].cpu().numpy()
    mpath = nb_model.nb_params(means, roc, sc, al, be)[3]
    np.testing.assert_array_equal(path, mpath)
    assert set(np.unique(path)) == {0, 1, 2}


def test_many_gamma_poisson_samples_overflow_the_list(ctx):
    """More gamma-Poisson samples than the streaming kernel's list holds (a region takes 1/16 of its wave's samples):
    K3h then classifies every sample itself; a list that just fits takes the normal way.  Both
    bit-exact against the model."""
    from oracle import nb_model
    for frac, seed in ((0.6, 31), (0.012, 32), (0.045, 34)):
        # (0.045: every region's own list is large enough -- one sample in 16 --, the SEGMENTS are not -- one in 64 of the
        # matrix between them: the regions that find their segment full are redone whole, the others take their lists)
        means, roc, sc, al, be = synthetic(seed, 50, 1024, 900, heavy_frac=frac)
        path = nb_model.nb_params(means, roc, sc, al, be)[3]
        share = (path == 2).mean()
        assert (share > 0.3) if frac > 0.5 else ((0.002 < share < 1 / 64) if frac < 0.02 else (1 / 50 < share < 1 / 17))
        got = ctx.sample_counts(means, roc, sc, al, be, seed=seed).cpu().numpy()
        np.testing.assert_array_equal(got, nb_model.sample_counts(means, roc, sc, al, be, seed))
        assert ctx.last_list()[3] == (frac > 0.02)
    # only the second of four gene tiles is that dense: its regions are redone whole, the others take their lists
    means, roc, sc, al, be = synthetic(33, 50, 1024, 900, heavy_frac=0.012)
    means[:, 256:512][:, np.random.default_rng(1).random(256) < 0.5] *= 200.0
    got = ctx.sample_counts(means, roc, sc, al, be, seed=33).cpu().numpy()
    np.testing.assert_array_equal(got, nb_model.sample_counts(means, roc, sc, al, be, 33))
    cells, genes, total, overflowed = ctx.last_list()
    assert overflowed and ((genes < 256) | (genes >= 512)).sum() > 100


def test_long_inversion_walks_bit_exact(ctx):
    """Means of 20-100 with theta <= 24 are drawn by inversion (walks of a hundred terms and more,
    results far later than their row's store): bit-exact, and the class is the model's."""
    from oracle import nb_model
    rng = np.random.default_rng(8)
    G, rows, N = 512, 16, 700
    means = np.exp(rng.uniform(np.log(15), np.log(110), (rows, G))).astype(np.float32)
    roc = rng.integers(0, rows, N).astype(np.int32)
    sc = np.exp(rng.normal(0, 0.3, N))
    al = rng.uniform(0.0, 0.15, G)
    be = rng.uniform(1.5, 6.0, G)
    want = nb_model.sample_counts(means, roc, sc, al, be, 5)
    path = nb_model.nb_params(means, roc, sc, al, be)[3]
    assert 0.3 < (path == 1).mean() < 0.95 and (path == 2).any()
    got = ctx.sample_counts(means, roc, sc, al, be, seed=5).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert want[path == 1].max() > 150


def test_domain_errors_like_scipy(ctx):
    """simulation.py:647-648: an exact-zero mean, or alpha*m + beta < 1, raises ValueError;
    alpha = 0, beta = 1 silently yields zeros."""
    means = np.ones((2, 8), np.float32)
    means[1, 3] = 0.0
    roc = np.array([0, 1], np.int32)
    sc = np.ones(2)
    with pytest.raises(ValueError):
        ctx.sample_counts(means, roc, sc, np.full(8, 0.2), np.full(8, 2.0), seed=1)
    ok = ctx.sample_counts(means, roc[:1], sc[:1], np.full(8, 0.2), np.full(8, 2.0), seed=1)
    assert ok.shape == (1, 8)
    with pytest.raises(ValueError):
        ctx.sample_counts(means, roc[:1], sc[:1], np.full(8, 0.0), np.full(8, 0.5), seed=1)
    z = ctx.sample_counts(means, roc[:1], sc[:1], np.zeros(8), np.ones(8), seed=1)
    assert int(z.sum()) == 0
    # unchecked mode never raises and writes 0 for the offending entries
    u = ctx.sample_counts(means, roc, sc, np.full(8, 0.2), np.full(8, 2.0), seed=1, check_domain=False)
    assert int(u[1, 3]) == 0


def test_deferred_domain_check_and_cached_row_flags(ctx):
    """check_domain="deferred": the same verdict as the synchronous check, but the call does not wait for it --
    ``domain_status()`` raises it (once) later.  With a ``means_token`` the per-row flags of the mean tensor are
    kept between calls: the same token vouches that the tensor is unchanged, a new token makes the next call rescan."""
    import torch
    from prosstt_amd import _native
    means = torch.ones((3, 8), dtype=torch.float32, device=ctx.torch_device)
    roc = np.array([0, 1, 2], np.int32)
    sc = np.ones(3)
    al, be = np.full(8, 0.2), np.full(8, 2.0)
    ctx.sample_counts(means, roc, sc, al, be, seed=1, check_domain="deferred", means_token="v1")
    ctx.domain_status()                                    # nothing to report
    means[2, 5] = 0.0                                      # edited in place ...
    ctx.sample_counts(means, roc, sc, al, be, seed=1, check_domain="deferred", means_token="v2")     # ... and announced
    ctx.sample_counts(means, roc[:2], sc[:2], al, be, seed=2, check_domain="deferred", means_token="v2")
    with pytest.raises(ValueError):
        ctx.domain_status()
    ctx.domain_status()                                    # the verdict was cleared when it was read
    # only the rows a call uses count
    ctx.sample_counts(means, roc[:2], sc[:2], al, be, seed=2, check_domain="deferred", means_token="v2")
    ctx.domain_status()
    # a scaling that is not > 0, a gene with beta < 1 (full per-sample pass), a row index outside the tensor
    ctx.sample_counts(means, roc[:2], np.array([1.0, 0.0]), al, be, seed=2, check_domain="deferred", means_token="v2")
    with pytest.raises(ValueError):
        ctx.domain_status()
    bad_beta = be.copy(); bad_beta[3] = 0.5
    ctx.sample_counts(means, roc[:2], sc[:2], np.zeros(8), bad_beta, seed=2, check_domain="deferred", means_token="v2")
    with pytest.raises(ValueError):
        ctx.domain_status()
    ctx.sample_counts(means, roc[:2], sc[:2], al, be, seed=2, check_domain="deferred", means_token="v2")
    ctx.domain_status()                                    # the request for the full pass does not outlive its call
    ctx.sample_counts(means, np.array([0, 7], np.int32), sc[:2], al, be, seed=2, check_domain="deferred", means_token="v2")
    with pytest.raises(_native.NativeError):
        ctx.domain_status()
    # without a token nothing is reused: an in-place edit is seen by the next checked call
    means[2, 5] = 1.0
    ctx.sample_counts(means, roc, sc, al, be, seed=3, check_domain=True)
    means[0, 0] = float("nan")
    with pytest.raises(ValueError):
        ctx.sample_counts(means, roc, sc, al, be, seed=3, check_domain=True)


@pytest.mark.parametrize("m0,theta0", [(101.0, 14.8), (136.0, 23.2)])
def test_corner_of_the_inversion_class_long_walks(ctx, m0, theta0):
    """5e7 samples at the corner of the inversion class -- PRNB-6's (theta about 23.2, mean about 136, tail ratio 0.959)
    and PRNB-6's (theta about 14.8, mean about 101): the longest walks the class has (hundreds of terms; round 3's
    definition let such walks run past the 1/k table).  Counts equal the model's one for one; nothing above the
    walk's end; first two moments hold."""
    from oracle import nb_model
    rng = np.random.default_rng(8)
    G, N = 20000, 2500
    means = (m0 * np.exp(rng.normal(0, 0.03, (4, G)))).astype(np.float32)
    roc = rng.integers(0, 4, N).astype(np.int32)
    sc = np.exp(rng.normal(0, 0.02, N))
    al = np.full(G, (theta0 - 1.0) / m0)
    be = np.full(G, 2.0)
    got = ctx.sample_counts(means, roc, sc, al, be, seed=2024).cpu().numpy()
    want = nb_model.sample_counts(means, roc, sc, al, be, 2024)
    np.testing.assert_array_equal(got, want)
    path = nb_model.nb_params(means[:, :64], roc[:50], sc[:50], al[:64], be[:64])[3]
    assert (path == 1).mean() > 0.5                      # the inversion class, mostly (t is just under 19 around the mean)
    mu = means[roc].astype(np.float64) * sc[:, None]
    var = al * mu * mu + be * mu
    assert abs(got.sum() / mu.sum() - 1) < 5 * np.sqrt(var.sum()) / mu.sum()
    assert abs(((got - mu) ** 2).sum() / var.sum() - 1) < 0.01
    assert got.max() < 3000 and got.min() >= 0


def test_empty_inputs(ctx):
    means = np.ones((3, 16), np.float32)
    out = ctx.sample_counts(means, np.zeros(0, np.int32), np.zeros(0), np.full(16, 0.2), np.full(16, 2.0), seed=3)
    assert tuple(out.shape) == (0, 16)


def test_law_against_scipy_tables(ctx):
    """Every parameter set of fixture g7: 2e6 device draws vs the analytic pmf."""
    from scipy import stats
    g = load_golden("g7_nb_tables")
    par, pmf = g["params"], g["pmf"]
    n = 2_000_000
    pvals = []
    for i, (m, a, b, r, p, mean, var) in enumerate(par):
        means = np.full((1, 4), m, np.float32)
        x = ctx.sample_counts(means, np.zeros(n // 4, np.int32), np.ones(n // 4), np.full(4, a), np.full(4, b),
                              seed=4242 + i, check_domain=False).cpu().numpy().reshape(-1)
        m32 = float(np.float32(m))
        assert abs(x.mean() - m32) < 6 * np.sqrt(var / n) + 1e-6 * m32
        kmax = pmf.shape[1]
        cnt = np.bincount(np.minimum(x, kmax - 1), minlength=kmax).astype(float)
        exp = pmf[i] * n
        exp[-1] += n * max(0.0, 1 - pmf[i].sum())
        obs_b, exp_b, co, ce = [], [], 0.0, 0.0
        for k in range(kmax):
            co += cnt[k]
            ce += exp[k]
            if ce >= 50:
                obs_b.append(co); exp_b.append(ce); co = ce = 0.0
        obs_b[-1] += co
        exp_b[-1] += ce
        obs_b, exp_b = np.array(obs_b), np.array(exp_b)
        chi2 = ((obs_b - exp_b) ** 2 / exp_b).sum()
        pvals.append(stats.chi2.sf(chi2, len(obs_b) - 1))
    assert min(pvals) > 1e-4 / len(pvals), pvals     # Bonferroni


def test_chunking_offset_and_determinism(ctx):
    """Results are a pure function of (inputs, seed, global cell index): sampling cells
    [0,N) in one call equals sampling them in ragged chunks with cell_offset."""
    import torch
    means, roc, sc, al, be = synthetic(21, 400, 5000, 5000)
    full = ctx.sample_counts(means, roc, sc, al, be, seed=77)
    again = ctx.sample_counts(means, roc, sc, al, be, seed=77)
    assert torch.equal(full, again)
    other = ctx.sample_counts(means, roc, sc, al, be, seed=78)
    assert not torch.equal(full, other)
    parts, start = [], 0
    for size in (1, 15, 16, 17, 1000, 3951):
        sl = slice(start, start + size)
        parts.append(ctx.sample_counts(means, roc[sl], sc[sl], al, be, seed=77, cell_offset=start))
        start += size
    assert start == 5000
    assert torch.equal(full, torch.cat(parts))
    # strided output (ld_out > G) lands in the same values
    wide = torch.zeros((5000, 5008), dtype=torch.int32, device=full.device)
    ctx.sample_counts(means, roc, sc, al, be, seed=77, out=wide[:, :5000])
    assert torch.equal(wide[:, :5000], full) and int(wide[:, 5000:].abs().sum()) == 0
    # first-moment check at this size
    mu = torch.as_tensor(means, device=full.device)[torch.as_tensor(roc, device=full.device).long()] \
        * torch.as_tensor(sc, device=full.device, dtype=torch.float32)[:, None]
    ratio = float(full.sum(dtype=torch.float64) / mu.sum(dtype=torch.float64))
    assert abs(ratio - 1) < 2e-3


@pytest.mark.parametrize("G", [20000, 20001])      # 16-byte rows / the scalar-layout path (G % 4 != 0)
def test_one_big_call_equals_its_small_chunks(ctx, G):
    """A call with full-length strips and at least 1 GiB of counts runs the stream kernel's BIG instantiation
    (raised issue priority in stages 2 and 3, system-scope row stores: k3_stream.h); the same cells in chunks under
    that size run the other one.  Counts are a pure function of (inputs, seed, global cell index): equal, bit for bit."""
    import torch
    N = 13500                                        # N * G * 4 = 1.08e9 bytes >= 2^30
    means, roc, sc, al, be = synthetic(33, 200, G, N, heavy_frac=0.002)
    full = ctx.sample_counts(means, roc, sc, al, be, seed=4321, cell_offset=5)
    start = 0
    for size in (6000, 6000, 1500):                  # <= 0.48e9 bytes each
        sl = slice(start, start + size)
        part = ctx.sample_counts(means, roc[sl], sc[sl], al, be, seed=4321, cell_offset=5 + start)
        assert torch.equal(full[sl], part)
        del part
        start += size
    assert start == N
    mu = torch.as_tensor(means, device=full.device)[torch.as_tensor(roc, device=full.device).long()] \
        * torch.as_tensor(sc, device=full.device, dtype=torch.float32)[:, None]
    assert abs(float(full.sum(dtype=torch.float64) / mu.sum(dtype=torch.float64)) - 1) < 2e-3


def test_row_index_outside_the_tensor_is_reported(ctx):
    """A row_of_cell entry outside the mean tensor: the checked mode reports it (EINVAL), nothing is read
    or written outside the tensor in either mode."""
    from prosstt_amd import _native
    means = np.ones((3, 16), np.float32)
    sc = np.ones(4)
    for bad in (3, -1, 2 ** 30):
        roc = np.array([0, 1, bad, 2], np.int32)
        with pytest.raises(_native.NativeError):
            ctx.sample_counts(means, roc, sc, np.full(16, 0.2), np.full(16, 2.0), seed=1)
        out = ctx.sample_counts(means, roc, sc, np.full(16, 0.2), np.full(16, 2.0), seed=1, check_domain=False)
        assert tuple(out.shape) == (4, 16)
        # a gene with beta < 1 makes the checked mode run its full per-sample domain pass: that pass, nb_params and the
        # second kernel must not follow the bad index either
        beta = np.full(16, 2.0)
        beta[5] = 0.5
        with pytest.raises((_native.NativeError, ValueError)):
            ctx.sample_counts(means, roc, sc, np.full(16, 0.2), beta, seed=1)
        mu, p, r, path = ctx.nb_params(means, roc, sc, np.full(16, 0.2), beta)
        assert tuple(mu.shape) == (4, 16)
    # a mean tensor without rows cannot serve any cell
    with pytest.raises(_native.NativeError):
        ctx.sample_counts(np.ones((0, 16), np.float32), np.zeros(2, np.int32), np.ones(2), np.full(16, 0.2), np.full(16, 2.0),
                          seed=1, check_domain=False)
