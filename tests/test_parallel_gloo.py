"""
CPU tests of the N>1 path: world_size-2 and -3 gloo processes run the branch sharding, the plan
broadcast, sample_density_sharded (lock-step host draws, replica digest check) and the chunked
row gather of prosstt_amd.parallel.  The device sampler itself cannot run here; a stand-in that
writes f(global cell id, gene) plays its role, which is exactly the property the real kernel has
(counts keyed by global cell id), so the reassembly is checked value for value.  The real kernel
under 2/4/8 ranks' shards is covered on one GPU by tests/test_gpu_sharding.py.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from prosstt_amd import parallel, simulation as sim, sim_utils as sut
    from prosstt_amd.tree import Tree
    try:
        np.random.seed(100 + rank)                 # ranks deliberately disagree: the plan must come from rank 0
        t = Tree(topology=[[0, 1], [0, 2], [2, 3], [2, 4]], time={b: 10 + b for b in range(5)},
                 num_branches=5, branch_points=2, modules=3, G=6)
        N = 101
        if rank == 0:
            np.random.seed(7)
            plan = (*sim._density_plan(t, N), sut.calc_scalings(N), 42)
        else:
            plan = None
        pt, br, sc, seed = parallel.broadcast_plan(plan)
        mine, owner = parallel.shard_cells(br, rank, world_size)
        assert set(owner.values()) <= set(range(world_size)) and len(owner) == len(np.unique(br))
        # every cell of a branch lives on the branch's owner
        assert all(owner[int(b)] == rank for b in br[mine])
        rows = sim.cell_rows(t, pt[mine], br[mine])
        assert rows.dtype == np.int32 and len(rows) == len(mine)
        fake = (torch.as_tensor(mine)[:, None] * 1000 + torch.arange(6)[None, :]).to(torch.int32)
        full = parallel.gather_rows(fake, mine, N)
        sizes = [None] * world_size
        dist.all_gather_object(sizes, len(mine))
        assert sum(sizes) == N
        if rank == 0:
            want = (torch.arange(N)[:, None] * 1000 + torch.arange(6)[None, :]).to(torch.int32)
            assert torch.equal(full, want)
            np.save(os.path.join(tmpdir, "pt.npy"), pt)
        else:
            assert full is None
        dist.barrier()
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_greedy_balance_and_single_rank():
    from prosstt_amd import parallel
    rng = np.random.default_rng(0)
    br = rng.choice(np.arange(8), size=8000)
    for world in (1, 2, 4, 8):
        owner = parallel.assign_branches_to_ranks(br, world)
        load = np.zeros(world)
        for b, r in owner.items():
            load[r] += (br == b).sum()
        assert load.sum() == 8000 and load.max() <= 8000 / world * 1.15
        got = np.sort(np.concatenate([parallel.shard_cells(br, r, world)[0] for r in range(world)]))
        np.testing.assert_array_equal(got, np.arange(8000))
    labels = np.array(["A", "B", "A", "C", "A"])
    mine, owner = parallel.shard_cells(labels, 0, 2)
    assert owner["A"] == 0 and sorted(owner.values()) == [0, 1, 1]
    np.testing.assert_array_equal(mine, [0, 2, 4])


class _FakeContext:
    """Stand-in for device.Context on a box without a GPU: CPU tensors, and a 'sampler' that writes
    f(global cell id, gene, seed) -- the one property of the real kernel the sharding relies on."""
    def __init__(self):
        import torch
        self.torch_device = torch.device("cpu")
        self.calls = []

    def tensor(self, host, dtype):
        import torch
        return torch.as_tensor(np.ascontiguousarray(host)).to(dtype)

    def sample_counts(self, means, rows, sc, alpha, beta, seed=0, cell_index=None, check_domain=True, **kw):
        import torch
        assert len(rows) == len(sc) == len(cell_index) and means.shape[1] == len(alpha) == len(beta)
        self.calls.append(len(rows))
        idx = torch.as_tensor(np.asarray(cell_index), dtype=torch.int64)
        values = (idx[:, None] * 1000 + torch.arange(means.shape[1])[None, :] + seed % 7).to(torch.int32)
        if kw.get("out") is not None:
            kw["out"].copy_(values)
            return kw["out"]
        return values

    def domain_status(self):
        pass


def _worker_sharded(rank, world_size, port, tmpdir, same_seed):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from prosstt_amd import device, parallel
    from prosstt_amd.tree import Tree
    fake = _FakeContext()
    device.get_context = lambda *a, **k: fake
    try:
        G, N = 5, 211
        t = Tree(topology=[["A", "B"]], time={"A": 30, "B": 20}, num_branches=2, branch_points=0, modules=3, G=G)
        rng = np.random.default_rng(5 if same_seed else 5 + rank)
        t.means = {b: np.exp(rng.normal(size=(t.time[b], G))) for b in "AB"}
        np.random.seed(77 if same_seed else 77 + rank)
        if not same_seed:
            # different trees on the ranks: strict mode must notice
            try:
                parallel.sample_density_sharded(t, N, seed=3)
                raise AssertionError("replicas disagree and nobody noticed")
            except RuntimeError as e:
                assert "different trees" in str(e)
            dist.barrier()
            open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
            return
        counts, mine, pt, br, sc = parallel.sample_density_sharded(t, N, seed=3)
        # two branches, three ranks: somebody owns nothing, and the pieces add up to the plan
        sizes = [None] * world_size
        dist.all_gather_object(sizes, len(mine))
        assert sum(sizes) == N and 0 in sizes and len(set(sizes)) == 3
        assert tuple(counts.shape) == (len(mine), G)
        # lock-step: every rank consumed the same draws, so the next variate is the same everywhere
        nxt = [None] * world_size
        dist.all_gather_object(nxt, float(np.random.random()))
        assert len(set(nxt)) == 1
        plans = [None] * world_size
        dist.all_gather_object(plans, parallel._digest(pt, np.array([str(b) for b in br]), sc))
        assert len(set(plans)) == 1
        # the gather: chunks smaller than every shard, all senders at once, an empty sender among them
        full = parallel.gather_rows(counts, mine, N, chunk_rows=16)
        want = (torch.arange(N)[:, None] * 1000 + torch.arange(G)[None, :] + 3).to(torch.int32)
        if rank == 0:
            assert torch.equal(full, want)
        else:
            assert full is None
        other = parallel.gather_rows(counts, mine, N, dst=1, chunk_rows=1000)      # another root, one round
        assert (other is not None) == (rank == 1)
        # nothing but rows exchanged: every rank derives every shard's cell indices from the broadcast plan
        layout = parallel.shards_in_presentation_order(t, pt, br, world_size)
        assert np.array_equal(layout[rank], mine)
        known = parallel.gather_rows(counts, mine, N, chunk_rows=16, index_of_rank=layout)
        assert torch.equal(known, want) if rank == 0 else known is None
        # (order="plan": the rank's cells in ascending plan position, as until round 4)
        np.random.seed(77)
        c_plan, mine_plan, _, _, _ = parallel.sample_density_sharded(t, N, seed=3, order="plan")
        assert np.array_equal(mine_plan, np.sort(mine)) and (len(mine) < 2 or (np.diff(mine_plan) > 0).all())
        again = parallel.gather_rows(c_plan, mine_plan, N)
        assert torch.equal(again, want) if rank == 0 else again is None
        # the same gather assembled in host memory on the root (a result too large for one device)
        on_host = parallel.gather_rows(counts, mine, N, chunk_rows=16, to_host=True)
        if rank == 0:
            assert isinstance(on_host, np.ndarray) and np.array_equal(on_host, want.numpy())
        else:
            assert on_host is None
        # sampling and exchange as ONE pipeline (chunk c travels while chunk c + 1 is sampled): the same matrix.
        # order="shard": every chunk is received straight into its final rows, the permutation comes back
        np.random.seed(77)
        before = len(fake.calls)
        piped, cell_of_row, pt3, br3, sc3 = parallel.sample_and_gather(t, N, seed=3, chunk_cells=16)
        assert np.array_equal(pt3, pt) and np.array_equal(sc3, sc)
        assert np.array_equal(np.sort(cell_of_row), np.arange(N))
        assert len(fake.calls) - before == -(-len(mine) // 16) and all(c <= 16 for c in fake.calls[before:])    # sampled chunk by chunk
        # (every rank knows the layout: its own cells sit where it would expect them)
        at = int(np.nonzero(cell_of_row == mine[0])[0][0]) if len(mine) else 0
        assert np.array_equal(cell_of_row[at:at + len(mine)], mine)
        if rank == 0:
            assert torch.equal(piped, want[torch.as_tensor(cell_of_row)])
            assert torch.equal(piped, parallel.gather_rows(counts, mine, N)[torch.as_tensor(cell_of_row)])
        else:
            assert piped is None and parallel.gather_rows(counts, mine, N) is None
        # order="plan": rows in plan order (staged and scattered on the root, one round behind the transfers); another root
        np.random.seed(77)
        piped, none, _, _, _ = parallel.sample_and_gather(t, N, seed=3, chunk_cells=1000, order="plan", dst=2)
        assert none is None and (piped is not None) == (rank == 2)
        if rank == 2:
            assert torch.equal(piped, want)
        np.random.seed(77)
        piped, _, _, _, _ = parallel.sample_and_gather(t, N, seed=3, chunk_cells=7, order="plan")
        if rank == 0:
            assert torch.equal(piped, want)
        dist.barrier()
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("same_seed", [True, False])
def test_three_ranks_sharded_sampling_and_gather(tmp_path, same_seed):
    """sample_density_sharded + gather_rows on 3 gloo ranks with unequal shards and an empty rank;
    ranks seeded alike stay in lock-step, ranks holding different trees are refused in strict mode."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_sharded, args=(3, port, str(tmp_path), same_seed), nprocs=3, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(3))


def _worker_subgroup(rank, world_size, port, tmpdir):
    """Ranks 1 and 2 of a 3-rank world form a sub-group: broadcast_plan from the group's rank 0
    (global rank 1) with string labels, gather_rows to the group's rank 1 (global rank 2)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from prosstt_amd import parallel
    try:
        sub = dist.new_group([1, 2])               # every rank of the world takes part in creating it
        if rank > 0:
            N, G = 57, 4
            if dist.get_rank(sub) == 0:
                rng = np.random.default_rng(3)
                plan = (rng.integers(0, 90, N), rng.choice(np.array(["root", "left", "right-branch"]), N),
                        rng.random(N), (1 << 63) + 12345)
            else:
                plan = None
            pt, br, sc, seed = parallel.broadcast_plan(plan, group=sub)
            rng = np.random.default_rng(3)
            np.testing.assert_array_equal(pt, rng.integers(0, 90, N))
            np.testing.assert_array_equal(br, rng.choice(np.array(["root", "left", "right-branch"]), N))
            np.testing.assert_array_equal(sc, rng.random(N))
            assert seed == (1 << 63) + 12345 and pt.dtype == np.int64 and sc.dtype == np.float64
            mine, _ = parallel.shard_cells(br, dist.get_rank(sub), 2)
            fake = (torch.as_tensor(mine)[:, None] * 10 + torch.arange(G)[None, :]).to(torch.int32)
            full = parallel.gather_rows(fake, mine, N, group=sub, dst=1)      # default chunking: a byte budget
            if dist.get_rank(sub) == 1:
                assert torch.equal(full, (torch.arange(N)[:, None] * 10 + torch.arange(G)[None, :]).to(torch.int32))
            else:
                assert full is None
            full = parallel.gather_rows(fake, mine, N, group=sub, dst=0, chunk_bytes=64)   # 4 rows per round
            assert (full is not None) == (dist.get_rank(sub) == 0)
        dist.barrier()
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_plan_broadcast_and_gather_inside_a_subgroup(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_subgroup, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(3))


class _FakeLineageContext(_FakeContext):
    """numpy stand-ins for the lineage kernels (K2a attempt: max of programs@H and per-sibling counts of
    genes with negative Pearson r over the common steps; K2b commit; gene_max; means_from_rel), on whatever
    gene range H covers -- which is all the sharded lineage needs from the device."""

    def lineage_attempt_batch(self, programs, H, sib_programs=()):
        Hn = H.numpy()
        tops = np.array([np.max(p @ Hn) for p in programs])
        counts = np.zeros((len(programs), len(sib_programs)), np.int64)
        for b, p in enumerate(programs):
            for j, s in enumerate(sib_programs):
                c = min(len(p), len(s))
                x, y = p[:c] @ Hn, np.asarray(s)[:c] @ Hn
                cov = ((x - x.mean(axis=0)) * (y - y.mean(axis=0))).sum(axis=0)
                counts[b, j] = int((cov < 0).sum())
        return tops, counts

    def lineage_commit(self, programs, H, rel_out=None, gene_max=None):
        import torch
        rel = torch.as_tensor(np.asarray(programs) @ H.numpy())
        rel_out.copy_(rel)
        torch.maximum(gene_max, rel.max(dim=0).values, out=gene_max)

    def gene_max(self, rel, gmax):
        import torch
        return torch.maximum(gmax, rel.max(dim=0).values, out=gmax) if rel.shape[0] else gmax

    def means_from_rel(self, rel, base, out=None):
        import torch
        return (torch.exp(rel) * base[None, :]).to(torch.float32)


def _lineage_case():
    from prosstt_amd.tree import Tree
    return Tree(topology=[["A", "B"], ["A", "C"], ["C", "D"], ["C", "E"], ["B", "F"]],
                time={"A": 12, "B": 9, "C": 14, "D": 7, "E": 10, "F": 8}, num_branches=6, branch_points=2, modules=4, G=37)


def _worker_lineage(rank, world_size, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from prosstt_amd import device, parallel, simulation as sim, sim_utils as sut
    fake = _FakeLineageContext()
    device.get_context = lambda *a, **k: fake
    try:
        # the single-process lineage (every rank computes it for itself as the yardstick)
        np.random.seed(31)
        ref_tree = _lineage_case()
        ref_stats = []
        ref_rel, ref_prog, ref_H = sim.simulate_lineage(ref_tree, a=0.3, intra_branch_tol=0, inter_branch_tol=0.25,
                                                        rel_exp_cutoff=6, stats=ref_stats)
        ref_after = np.random.random()
        ref_base = None
        np.random.seed(32)
        ref_base = sut.simulate_base_gene_exp(ref_tree, ref_rel)
        ref_tree.add_genes(ref_rel, ref_base)
        # the same by all ranks together
        np.random.seed(31)
        t = _lineage_case()
        stats = []
        rel, prog, H = parallel.simulate_lineage_sharded(t, 6, 0, 0.25, a=0.3, stats=stats)
        assert np.random.random() == ref_after                        # numpy's stream: same position as the single process
        assert np.array_equal(H, ref_H) and list(prog.keys()) == list(ref_prog.keys())
        assert all(np.array_equal(prog[b], ref_prog[b]) for b in prog.keys())
        assert len(stats) == len(ref_stats) > len(t.branches)         # some attempts were rejected, the same ones
        for (b0, top0, c0), (b1, top1, c1) in zip(stats, ref_stats):
            assert b0 == b1 and c0 == c1 and abs(top0 - top1) <= 1e-12 * max(1.0, abs(top1))
        owner = parallel.assign_branches_by_density(t, world_size)
        held = [b for b in t.branches if owner[b] == rank]
        assert t.resident_branches() == held and sorted(rel.keys()) == sorted(held)
        assert t._lineage["rel"].shape[0] == sum(t.time[b] for b in held)           # nobody holds the whole tree
        for b in held:
            np.testing.assert_allclose(rel[b], ref_rel[b], rtol=1e-13, atol=1e-13)
        np.random.seed(32)
        base = sut.simulate_base_gene_exp(t, rel)                     # the all-reduced per-gene maximum
        np.testing.assert_array_equal(base, ref_base)
        t.add_genes(rel, base)
        assert sorted(t.means.keys()) == sorted(held)
        for b in held:
            np.testing.assert_allclose(t.means[b], ref_tree.means[b], rtol=1e-6)
        # sampling on the sharded tree: every cell on the rank that holds its branch, gathered in plan order
        np.random.seed(33)
        counts, mine, pt, br, sc = parallel.sample_density_sharded(t, 150, seed=9)
        assert all(owner[str(b)] == rank for b in br[mine])
        full = parallel.gather_rows(counts, mine, 150)
        if rank == 0:
            want = (torch.arange(150)[:, None] * 1000 + torch.arange(t.G)[None, :] + 9 % 7).to(torch.int32)
            assert torch.equal(full, want)
        sizes = [None] * world_size
        dist.all_gather_object(sizes, (len(held), len(mine)))
        assert sum(s[0] for s in sizes) == len(t.branches) and sum(s[1] for s in sizes) == 150
        dist.barrier()
        open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 3])
def test_lineage_sharded_by_genes_equals_the_single_process_lineage(tmp_path, world_size):
    """parallel.simulate_lineage_sharded on 2 and 3 gloo ranks: programs, attempt records (maxima, anticorrelated
    counts), numpy stream position, base expression and the owned rows equal the single-process run; no rank
    holds more than its own branches."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_lineage, args=(world_size, port, str(tmp_path)), nprocs=world_size, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world_size))
