"""
-m gpu: the branch-sharded multi-GPU path gives the single-GPU matrix.  One GPU plays every
rank in turn: each "rank" samples the cells of the branches it owns, keyed by their positions in
the global plan (cell_index), and the shards reassemble to exactly the unsharded result --
for 2, 4 and 8 ranks.  (The collective itself is covered by the gloo test on CPU.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_shards_reassemble_to_the_single_gpu_result():
    import torch
    from prosstt_amd import device, parallel, simulation as sim, workloads
    ctx = device.get_context()
    work = workloads.build("C3", G=1024)
    N = 3000
    pt, br, sc, rows = work.plan(N)
    means = work.tree.device_means()
    full = ctx.sample_counts(means, rows, sc, work.alpha, work.beta, seed=77)
    for world in (2, 4, 8):
        out = torch.full_like(full, -1)
        sizes = []
        for rank in range(world):
            mine, owner = parallel.shard_cells(br, rank, world)
            sizes.append(len(mine))
            assert all(owner[int(b)] == rank for b in np.unique(br[mine]))
            part = ctx.sample_counts(means, sim.cell_rows(work.tree, pt[mine], br[mine]), sc[mine],
                                     work.alpha, work.beta, seed=77, cell_index=mine)
            out[torch.as_tensor(mine, device=out.device)] = part
        assert sum(sizes) == N and max(sizes) <= 1.3 * N / world + 50
        assert torch.equal(out, full)


def test_sample_density_sharded_single_process_equals_sample_density():
    from prosstt_amd import parallel, simulation as sim, workloads
    work = workloads.build("C2", G=512)
    np.random.seed(5)
    X, pt, br, sc = sim.sample_density(work.tree, 700, alpha=work.alpha, beta=work.beta, seed=9, out="torch", order="plan")
    np.random.seed(5)
    counts, idx, pt2, br2, sc2 = parallel.sample_density_sharded(work.tree, 700, alpha=work.alpha, beta=work.beta, seed=9)
    np.testing.assert_array_equal(pt, pt2)
    np.testing.assert_array_equal(sc, sc2)
    np.testing.assert_array_equal(np.sort(idx), np.arange(700))      # (row i of `counts` is cell idx[i]: grouped by mean-tensor row)
    full = parallel.gather_rows(counts, idx, 700)
    import torch
    assert torch.equal(full, X)


def test_sample_and_gather_single_process_equals_sample_density():
    """parallel.sample_and_gather without a process group: the pipeline's chunked sampling into final rows, both orders."""
    import torch
    from prosstt_amd import parallel, simulation as sim, workloads
    work = workloads.build("C2", G=512)
    np.random.seed(5)
    X, pt, br, sc = sim.sample_density(work.tree, 700, alpha=work.alpha, beta=work.beta, seed=9, out="torch", order="plan")
    np.random.seed(5)
    full, cell_of_row, pt2, br2, sc2 = parallel.sample_and_gather(work.tree, 700, alpha=work.alpha, beta=work.beta, seed=9, chunk_cells=128)
    np.testing.assert_array_equal(pt, pt2)
    np.testing.assert_array_equal(np.sort(cell_of_row), np.arange(700))
    assert torch.equal(full, X[torch.as_tensor(cell_of_row, device=X.device)])
    np.random.seed(5)
    planned, none, _, _, _ = parallel.sample_and_gather(work.tree, 700, alpha=work.alpha, beta=work.beta, seed=9, chunk_cells=300, order="plan")
    assert none is None and torch.equal(planned, X)
