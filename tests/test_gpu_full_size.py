"""
-m gpu: BASELINE.json's headline configuration (C3: 8-branch tree, 50 000 cells x 20 000
genes) at FULL size, through size-independent properties -- the oracle cannot run 1e9
samples, so the full matrix is checked by
  * bit-exact agreement of sampled row blocks with the C model,
  * chunk/offset invariance (a row block recomputed alone, with cell_offset, equals the
    block inside the full launch),
  * sum(X)/sum(mu) -> 1 and the zero fraction against the closed form sum P0,
  * run-to-run determinism (checksum of the whole matrix).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c3_full_size_properties():
    import torch
    from prosstt_amd import device, workloads
    from oracle import nb_model
    ctx = device.get_context()
    work = workloads.build("C3")
    assert work.info["branches"] == 8 and work.info["rows"] == 400 and work.tree.G == 20000
    pt, br, sc, rows = work.plan()
    N, G = len(rows), work.tree.G
    assert N == 50000
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=20240)
    assert tuple(X.shape) == (N, G) and X.dtype == torch.int32
    checksum = int(X.sum(dtype=torch.int64)), int((X.to(torch.int64) * 2654435761 % 1000003).sum())
    # row blocks against the scalar model (ragged starts, spanning the strip boundaries at multiples of 64)
    host_means = means.cpu().numpy()
    for start, size in ((0, 3), (60, 10), (125, 7), (24571, 70), (49990, 10)):
        sl = slice(start, start + size)
        want = nb_model.sample_counts(host_means, rows[sl], sc[sl], work.alpha, work.beta, 20240, cell_offset=start)
        np.testing.assert_array_equal(X[sl].cpu().numpy(), want)
        alone = ctx.sample_counts(means, rows[sl], sc[sl], work.alpha, work.beta, seed=20240, cell_offset=start)
        assert torch.equal(alone, X[sl])
    # moments over the whole matrix
    mu_rows = means.double()                                           # (400, G)
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
    # determinism
    Y = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=20240)
    assert (int(Y.sum(dtype=torch.int64)), int((Y.to(torch.int64) * 2654435761 % 1000003).sum())) == checksum
    assert torch.equal(X[::997], Y[::997])
