"""
-m gpu: whole matrices, not samples of them.
  * C3 (50 000 x 20 000 = 1e9 counts, the headline workload), T32 (north_star's target shape: C4's 32-branch tree at C3's size,
    its cells presented grouped by mean-tensor row as simulation.draw_counts presents them), C4 (200 000 x 20 000) and C2 (5 000 x 5 000) compared with the
    scalar C model count for count: every class of sample, every late result and list entry of the full launch (the model runs on all host cores,
    block by block, so the host never holds more than one block of expected counts);
  * C5 at its full 1 000 000 cells x 30 000 genes on ONE GPU, the way eight GPUs would split it: the
    eight branch shards (cell_index = positions in the global plan) give, cell for cell, the row
    checksums of a plain pass over the plan in contiguous chunks.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _whole_matrix_vs_model(name, n_cells, block, grouped=False, seed=777):
    """grouped: the cells presented grouped by their row of the mean tensor, keyed by their position in the plan (what
    simulation.draw_counts does); everything below then runs on the matrix in that order."""
    import torch
    from prosstt_amd import device, workloads
    from oracle import nb_model
    ctx = device.get_context()
    work = workloads.build(name, verbose=True)
    pt, br, sc, rows = work.plan(n_cells)
    assert len(rows) == n_cells == work.cfg["N"]
    means = work.tree.device_means()
    cell = np.arange(n_cells, dtype=np.int64)
    if grouped:
        cell = device.plan_order(rows, means.shape[0]).astype(np.int64)
        rows, sc = rows[cell], sc[cell]
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=seed, cell_index=cell if grouped else None)
    host_means = means.cpu().numpy()
    differing = 0
    for lo in range(0, n_cells, block):
        sl = slice(lo, min(lo + block, n_cells))
        want = nb_model.sample_counts(host_means, rows[sl], sc[sl], work.alpha, work.beta, seed, cell_index=cell[sl])
        got = X[sl].cpu().numpy()
        differing += int((got != want).sum())
    assert differing == 0, "%d of %d counts differ from the model" % (differing, X.numel())
    # and the reference's law at this scale: first two moments of the whole matrix against mean mu and variance
    # v = alpha*mu^2 + beta*mu (count_model.py:131-161), in binary64 on the device, each against its own sampling
    # error: Var[X] = v, Var[(X - mu)^2] = kappa4 + 2 v^2 with the NB's kappa4 = v (1 + 6 theta + 6 theta^2),
    # theta = v/mu - 1 (a few high-expression genes carry most of both sums, so the errors are not tiny)
    d_rows = torch.as_tensor(rows, device=X.device).long()
    d_sc = torch.as_tensor(sc, device=X.device)
    d_al = torch.as_tensor(work.alpha, device=X.device)
    d_be = torch.as_tensor(work.beta, device=X.device)
    sx = smu = sdev = svar = sk = chi = chi_var = chi_n = 0.0
    h_n = h_x = h_mu = h_v = h_chi = h_chi_var = h_zero = h_p0 = h_p0_var = 0.0
    obs_k, exp_k, var_k = (torch.zeros(10, dtype=torch.float64, device=X.device) for _ in range(3))
    for lo in range(0, n_cells, 2048):                # bounded temporaries
        sl = slice(lo, min(lo + 2048, n_cells))
        mu = means[d_rows[sl]].double() * d_sc[sl, None]
        x = X[sl].double()
        v = d_al * mu * mu + d_be * mu
        theta = v / mu - 1.0
        sx += float(x.sum()); smu += float(mu.sum())
        sdev += float(((x - mu) ** 2).sum()); svar += float(v.sum())
        k4 = v * (1.0 + 6.0 * theta + 6.0 * theta * theta)
        sk += float((k4 + 2.0 * v * v).sum())
        # the same per sample in units of its own variance, over the samples with mu >= 0.05 (below that a sample's
        # term is almost always ~0 and rarely 1/v: the sum would hang on a handful of them): every sample counts
        # alike (mean 1, variance 2 + kappa4/v^2)
        big = mu >= 0.05
        r = mu / theta
        # the gamma-Poisson class on its own (theta > 24 or -log P(0) > 19: 0.1-0.2 % of the samples, drawn by K3h)
        hv = (theta > 24.0) | (r * torch.log1p(theta) > 19.0)
        h_n += float(hv.sum()); h_x += float(x[hv].sum()); h_mu += float(mu[hv].sum()); h_v += float(v[hv].sum())
        h_chi += float(((x - mu) ** 2 / v)[hv].sum()); h_chi_var += float((2.0 + k4 / (v * v))[hv].sum())
        h_zero += float((x[hv] == 0).sum())
        p0h = torch.exp(-r * torch.log1p(theta))[hv]
        h_p0 += float(p0h.sum()); h_p0_var += float((p0h * (1.0 - p0h)).sum())
        pk = torch.exp(-r * torch.log1p(theta))
        ratio = theta / (1.0 + theta)
        for k in range(10):
            obs_k[k] += (X[sl] == k).sum()
            exp_k[k] += pk.sum(); var_k[k] += (pk * (1.0 - pk)).sum()
            pk = pk * (r + k) / (k + 1.0) * ratio
        chi += float(((x - mu) ** 2 / v)[big].sum()); chi_var += float((2.0 + k4 / (v * v))[big].sum()); chi_n += float(big.sum())
    # and the histogram of the whole matrix against the pmf itself: for k = 0..9 the number of samples equal to k
    # against the sum of P(X = k) over all samples (binary64 recurrence from P(X = 0) = (1+theta)^(-mu/theta)), within
    # its sampling error sqrt(sum p (1 - p)); at 1e9 samples that resolves 3e-5 of a probability
    zk = (obs_k - exp_k) / var_k.sqrt()
    assert float(zk.abs().max()) < 5, (zk.tolist(), (obs_k / exp_k).tolist())
    zh = ((h_x - h_mu) / h_v ** 0.5, (h_chi - h_n) / h_chi_var ** 0.5, (h_zero - h_p0) / max(h_p0_var, 1e-300) ** 0.5)
    assert h_n > 1e4 and max(abs(z) for z in zh) < 5, (h_n, zh)
    print("[%s] gamma-Poisson class alone, %.3g samples: sum(X)/sum(mu) = %.5f (z = %.2f), mean((X-mu)^2/v) = %.5f (z = %.2f), "
          "zeros observed/expected = %.5f (z = %.2f)" % (name, h_n, h_x / h_mu, zh[0], h_chi / h_n, zh[1], h_zero / max(h_p0, 1e-300), zh[2]))
    z1 = (sx - smu) / svar ** 0.5
    z2 = (sdev - svar) / sk ** 0.5
    n = chi_n
    z3 = (chi - n) / chi_var ** 0.5
    assert abs(z1) < 5 and abs(z2) < 5 and abs(z3) < 5, (z1, z2, z3, sx / smu, sdev / svar, chi / n)
    print("[%s] histogram k = 0..9, observed/expected - 1: %s; z: %s" % (name, ["%.1e" % v for v in (obs_k / exp_k - 1).tolist()],
                                                                        ["%.1f" % v for v in zk.tolist()]))
    print("[%s] %d x %d counts equal the model's; sum(X)/sum(mu) = %.5f (z = %.2f), sum((X-mu)^2)/sum(alpha mu^2 + beta mu) = %.5f "
          "(z = %.2f), mean((X-mu)^2/v) over the %.3g samples with mu >= 0.05 = %.6f (z = %.2f)"
          % (name, n_cells, work.tree.G, sx / smu, z1, sdev / svar, z2, n, chi / n, z3))


def test_c3_entire_matrix_equals_the_model():
    _whole_matrix_vs_model("C3", 50000, 5000)


def test_c2_entire_matrix_equals_the_model():
    _whole_matrix_vs_model("C2", 5000, 5000)


def test_t32_entire_matrix_equals_the_model_cells_grouped_by_row():
    """north_star's own target shape: the 32-branch tree at 50 000 x 20 000 (C4's tree, the headline's size), presented
    the way the drop-in API presents it."""
    _whole_matrix_vs_model("T32", 50000, 5000, grouped=True)


def test_c4_entire_matrix_equals_the_model():
    """200 000 x 20 000 = 4e9 counts of a 32-branch tree (twice C3's share of gamma-Poisson samples)."""
    _whole_matrix_vs_model("C4", 200000, 10000)


def _row_checksums(X):
    """Order-sensitive 64-bit checksum of every row (wraps modulo 2^64: fine for equality)."""
    import torch
    G = X.shape[1]
    w = (torch.arange(G, device=X.device, dtype=torch.int64) * 2654435761 + 12345) % 1000003
    out = torch.empty(X.shape[0], dtype=torch.int64, device=X.device)
    for lo in range(0, X.shape[0], 16384):            # bounded temporaries
        out[lo:lo + 16384] = (X[lo:lo + 16384].to(torch.int64) * w[None, :]).sum(dim=1)
    return out


def test_c5_full_size_eight_shards_on_one_gpu():
    import torch
    from prosstt_amd import device, parallel, simulation as sim, workloads
    ctx = device.get_context()
    work = workloads.build("C5", verbose=True)
    N, G = work.cfg["N"], work.tree.G
    assert (N, G, work.info["branches"]) == (1000000, 30000, 256)
    pt, br, sc, rows = work.plan(N)
    means = work.tree.device_means()
    # a plain pass over the plan, 125 000 contiguous cells at a time (15 GB of counts per chunk)
    whole = torch.empty(N, dtype=torch.int64, device=ctx.torch_device)
    total = 0
    chunk = 125000
    for lo in range(0, N, chunk):
        X = ctx.sample_counts(means, rows[lo:lo + chunk], sc[lo:lo + chunk], work.alpha, work.beta, seed=31,
                              cell_offset=lo)
        whole[lo:lo + chunk] = _row_checksums(X)
        total += int(X.sum(dtype=torch.int64))
        del X
    mu_sum = float((means.double().sum(dim=1)[torch.as_tensor(rows, device=whole.device).long()]
                    * torch.as_tensor(sc, device=whole.device)).sum())
    assert abs(total / mu_sum - 1) < 1e-3
    # the eight ranks' shards: disjoint branch sets, cells keyed by their position in the global plan
    seen = 0
    for rank in range(8):
        mine, owner = parallel.shard_cells(br, rank, 8)
        assert 0.8 * N / 8 < len(mine) < 1.2 * N / 8
        part = ctx.sample_counts(means, sim.cell_rows(work.tree, pt[mine], br[mine]), sc[mine], work.alpha,
                                 work.beta, seed=31, cell_index=mine)
        assert torch.equal(_row_checksums(part), whole[torch.as_tensor(mine, device=whole.device)])
        seen += len(mine)
        del part
    assert seen == N
    print("[C5] %d x %d: 8 shards == chunked pass; lineage attempts %d in %.2f s; sum(X)/sum(mu) = %.5f"
          % (N, G, work.info["attempts"], work.info["lineage_s"], total / mu_sum))


def test_every_listed_sample_has_a_reason():
    """The streaming kernel draws the inversion class itself and lists for the second kernel (K3h) only: the
    gamma-Poisson class, the few walks still running when their strip of 64 cells had nothing else to do (at most
    k3::kBail = 6 per wave), and walks past k = 248 (the row ring holds 8 bits).  On C3 the list of the last launch
    is read back (prosstt_amd_last_list): every entry must be one of those in the model, no sample of the
    gamma-Poisson class of a block of cells may be missing from it, and what K3h wrote is the model's count."""
    import torch
    from prosstt_amd import device, workloads
    from oracle import nb_model
    ctx = device.get_context()
    work = workloads.build("C3")
    N = 50000
    pt, br, sc, rows = work.plan(N)
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=424242)
    cells, genes, total, overflowed = ctx.last_list()
    assert not overflowed and total == len(cells)
    assert 2e5 < total < 0.01 * N * work.tree.G            # ~1.4e6 of 1e9 samples
    host_means = means.cpu().numpy()
    path, count = nb_model.sample_selected(host_means, rows, sc, work.alpha, work.beta, 424242, cells, genes)
    got = X[torch.as_tensor(cells, device=X.device), torch.as_tensor(genes.astype(np.int64), device=X.device)].cpu().numpy()
    np.testing.assert_array_equal(got, count)
    assert path.min() >= 1                                            # degenerate samples are never listed
    heavy = path == 2
    big = (path == 1) & (count > 248)
    leftover = ~(heavy | big)                                         # unfinished walks: they entered stage 3, so count >= 5
    waves = -(-N // 64) * -(-work.tree.G // 256)
    assert leftover.sum() <= 6 * waves and (count[leftover] >= 5).all(), \
        "%d listed samples without a reason" % int((leftover & (count < 5)).sum())
    assert heavy.sum() > 1e5
    print("[list] %d entries: gamma-Poisson %d, above 248: %d, unfinished at the end of their strip %d"
          % (total, heavy.sum(), big.sum(), leftover.sum()))
    # the other direction, on a block of cells: every sample of the gamma-Poisson class is on the list
    blk = np.arange(2000, 2300)
    bpath = nb_model.nb_params(host_means, rows[blk], sc[blk], work.alpha, work.beta)[3]
    bc, bg = np.nonzero(bpath == 2)
    in_blk = (cells >= 2000) & (cells < 2300)
    listed = set(zip(cells[in_blk].tolist(), genes[in_blk].tolist()))
    missing = [(int(c) + 2000, int(g)) for c, g in zip(bc, bg) if (int(c) + 2000, int(g)) not in listed]
    assert len(bc) > 100 and not missing, "%d samples of the gamma-Poisson class were not listed" % len(missing)
