"""
-m gpu: BASELINE.json's configurations at FULL size, through size-independent properties -- the
oracle cannot run 1e9 samples lane by lane in a test, so a full matrix is checked by
  * bit-exact agreement of sampled row blocks with the C model (a few rows at ragged starts, and one
    block of thousands of cells: the model runs on all host cores),
  * chunk/offset invariance (a row block recomputed alone, with cell_offset, equals the
    block inside the full launch),
  * sum(X)/sum(mu) -> 1 and the zero fraction against the closed form sum P0,
  * run-to-run determinism (checksum of the whole matrix).
C3 (8 branches, 50 000 x 20 000, the headline metric), C4 (32 branches, 200 000 x 20 000) and the
per-GPU share of C5 (256 branches, 30 000 genes: 125 000 of its 1 000 000 cells) go through the
product's own lineage stage first; the number of lineage attempts is reported.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _checksum(X):
    import torch
    return int(X.sum(dtype=torch.int64)), int((X.to(torch.int64) * 2654435761 % 1000003).sum())


def _full_size_properties(name, branches, rows_total, G, n_cells, blocks, big_block):
    import torch
    from prosstt_amd import device, workloads
    from oracle import nb_model
    ctx = device.get_context()
    work = workloads.build(name, verbose=True)
    assert work.info["branches"] == branches and work.info["rows"] == rows_total and work.tree.G == G
    pt, br, sc, rows = work.plan(n_cells)
    N = len(rows)
    assert N == n_cells
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=20240)
    assert tuple(X.shape) == (N, G) and X.dtype == torch.int32
    checksum = _checksum(X)
    # row blocks against the scalar model (ragged starts, spanning the strip boundaries at multiples of 64)
    host_means = means.cpu().numpy()
    for start, size in blocks:
        sl = slice(start, start + size)
        want = nb_model.sample_counts(host_means, rows[sl], sc[sl], work.alpha, work.beta, 20240, cell_offset=start)
        np.testing.assert_array_equal(X[sl].cpu().numpy(), want)
        alone = ctx.sample_counts(means, rows[sl], sc[sl], work.alpha, work.beta, seed=20240, cell_offset=start)
        assert torch.equal(alone, X[sl])
    # one large block: every class of sample, every margin of the hardware-math evaluation, 1e8 times over
    start, size = big_block
    sl = slice(start, start + size)
    want = nb_model.sample_counts(host_means, rows[sl], sc[sl], work.alpha, work.beta, 20240, cell_offset=start)
    got = X[sl].cpu().numpy()
    assert np.array_equal(got, want), "%d of %d counts differ from the model" % ((got != want).sum(), got.size)
    del want, got
    # moments over the whole matrix
    mu_rows = means.double()
    d_rows = torch.as_tensor(rows, device=X.device).long()
    d_sc = torch.as_tensor(sc, device=X.device)
    mu_sum = float((mu_rows.sum(dim=1)[d_rows] * d_sc).sum())
    assert abs(checksum[0] / mu_sum - 1) < 1e-3
    # closed-form zero fraction on a block of cells: P0 = (1+theta)^(-m/theta)
    blk = slice(1000, 1400)
    m = mu_rows[d_rows[blk]] * d_sc[blk, None]
    a = torch.as_tensor(work.alpha, device=X.device)
    theta = a * m + torch.as_tensor(work.beta, device=X.device) - 1
    p0 = torch.exp(-m / theta * torch.log1p(theta))
    zeros = float((X[blk] == 0).double().mean())
    assert abs(zeros - float(p0.mean())) < 4 * float(torch.sqrt((p0 * (1 - p0)).sum())) / p0.numel() + 1e-4
    del mu_rows, m, theta, p0
    # determinism
    Y = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=20240)
    assert _checksum(Y) == checksum
    assert torch.equal(X[::997], Y[::997])
    print("[%s] %d x %d: lineage attempts %d in %.2f s; sum(X)/sum(mu) = %.5f"
          % (name, N, G, work.info["attempts"], work.info["lineage_s"], checksum[0] / mu_sum))


def test_c3_full_size_properties():
    _full_size_properties("C3", 8, 400, 20000, 50000,
                          ((0, 3), (60, 10), (125, 7), (24571, 70), (49990, 10)), (30000, 6000))


def test_c4_full_size_properties():
    _full_size_properties("C4", 32, 1600, 20000, 200000,
                          ((0, 5), (63, 3), (100001, 66), (199995, 5)), (150000, 2500))


def test_c5_one_gpu_share_properties():
    """The share of one of 8 GPUs: 125 000 cells of the 256-branch tree, 30 000 genes."""
    _full_size_properties("C5", 256, 12800, 30000, 125000,
                          ((0, 5), (64, 2), (77777, 70), (124990, 10)), (60000, 1500))
