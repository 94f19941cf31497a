"""
-m gpu: whole matrices, not samples of them.
  * C3 (50 000 x 20 000 = 1e9 counts, the headline workload) and C2 (5 000 x 5 000) compared with the
    scalar C model count for count: every class of sample, every threshold margin of the hardware-math
    evaluation, every late result and list entry of the full launch (the model runs on all host cores,
    block by block, so the host never holds more than one block of expected counts);
  * C5 at its full 1 000 000 cells x 30 000 genes on ONE GPU, the way eight GPUs would split it: the
    eight branch shards (cell_index = positions in the global plan) give, cell for cell, the row
    checksums of a plain pass over the plan in contiguous chunks.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _whole_matrix_vs_model(name, n_cells, block):
    import torch
    from prosstt_amd import device, workloads
    from oracle import nb_model
    ctx = device.get_context()
    work = workloads.build(name, verbose=True)
    pt, br, sc, rows = work.plan(n_cells)
    assert len(rows) == n_cells == work.cfg["N"]
    means = work.tree.device_means()
    X = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=777)
    host_means = means.cpu().numpy()
    differing = 0
    for lo in range(0, n_cells, block):
        sl = slice(lo, min(lo + block, n_cells))
        want = nb_model.sample_counts(host_means, rows[sl], sc[sl], work.alpha, work.beta, 777, cell_offset=lo)
        got = X[sl].cpu().numpy()
        differing += int((got != want).sum())
    assert differing == 0, "%d of %d counts differ from the model" % (differing, X.numel())
    print("[%s] %d x %d counts equal the model's" % (name, n_cells, work.tree.G))


def test_c3_entire_matrix_equals_the_model():
    _whole_matrix_vs_model("C3", 50000, 5000)


def test_c2_entire_matrix_equals_the_model():
    _whole_matrix_vs_model("C2", 5000, 5000)


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
