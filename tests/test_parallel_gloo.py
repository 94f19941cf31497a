"""
CPU tests of the N>1 path: world_size-2 gloo processes run the branch sharding, the plan
broadcast and the row gather of prosstt_amd.parallel.  The device sampler itself cannot run
here; a stand-in that writes f(global cell id, gene) plays its role, which is exactly the
property the real kernel has (counts keyed by global cell id), so the reassembly is checked
value for value.
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
