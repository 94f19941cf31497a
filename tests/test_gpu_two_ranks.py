"""
-m gpu: the N > 1 path with the REAL kernel.  Two fresh child processes (gloo backend, both on
cuda:0 -- the test box has one GPU; the children are started before anything in them touches the GPU)
build the tree together (parallel.simulate_lineage_sharded: attempts on gene slices, one all-reduce per batch,
every rank keeps the branches it owns), run parallel.sample_density_sharded and parallel.gather_rows; rank 0
compares the gathered matrix with the single-process result (whole tree, same plan and seed).  The plan travels by tensor
broadcasts (the ranks are seeded differently on purpose: rank 0's plan must win).
"""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
from prosstt_amd import device, parallel, simulation as sim, sim_utils as sut, workloads
from prosstt_amd.tree import Tree

work = workloads.build("C2", G=1536)            # under the process group: the lineage is built sharded (gene slices, owned branches)
tree = work.tree
assert work.info["sharded"] and tree._branch_owner is not None
held = tree.resident_branches()
assert 0 < len(held) < len(tree.branches) and tree.device_means().shape[0] == sum(int(tree.time[b]) for b in held)
N = 2500
np.random.seed(1000 + rank)                      # the ranks' own streams differ: the plan must be rank 0's
counts, mine, pt, br, sc = parallel.sample_density_sharded(tree, N, alpha=work.alpha, beta=work.beta, seed=11)
assert counts.is_cuda and counts.dtype == torch.int32 and tuple(counts.shape) == (len(mine), tree.G)
sizes = [None] * world
dist.all_gather_object(sizes, len(mine))
assert sum(sizes) == N and min(sizes) > 0
full = parallel.gather_rows(counts, mine, N, chunk_rows=500)
# sampling and exchange as one pipeline: chunks of 300 cells, received straight into their final rows on rank 0
np.random.seed(1000 + rank)
piped, cell_of_row, pt_p, br_p, sc_p = parallel.sample_and_gather(tree, N, alpha=work.alpha, beta=work.beta, seed=11, chunk_cells=300)
np.random.seed(1000 + rank)
planned, _, _, _, _ = parallel.sample_and_gather(tree, N, alpha=work.alpha, beta=work.beta, seed=11, chunk_cells=700, order="plan")
assert np.array_equal(pt_p, pt) and np.array_equal(sc_p, sc) and np.array_equal(np.sort(cell_of_row), np.arange(N))
if rank == 0:
    np.random.seed(1000)
    pt0, br0 = sim._density_plan(tree, N)
    sc0 = sut.calc_scalings(N)
    assert np.array_equal(pt, pt0) and np.array_equal(br, br0) and np.array_equal(sc, sc0)
    ctx = device.get_context()
    whole = workloads.build("C2", G=1536, sharded=False).tree     # the single-process tree: every branch, same draws
    assert whole.resident_branches() == list(whole.branches)
    at, _ = whole.row_offsets()
    mine_at, _ = tree.row_offsets()
    for b in held:                                                 # the owned rows are the single-process rows
        assert torch.equal(tree.device_means()[mine_at[b]:mine_at[b] + int(tree.time[b])],
                           whole.device_means()[at[b]:at[b] + int(whole.time[b])])
    want = ctx.sample_counts(whole.device_means(), sim.cell_rows(whole, pt, br), sc, work.alpha, work.beta, seed=11)
    assert torch.equal(full, want), "gathered shards differ from the single-process matrix"
    assert torch.equal(piped, want[torch.as_tensor(cell_of_row, device=want.device)]), "pipelined gather (order='shard') differs"
    assert torch.equal(planned, want), "pipelined gather (order='plan') differs"
    assert int(want.sum()) > 0
    print("TWO_RANKS_OK", sizes, int(want.sum()), flush=True)
else:
    assert full is None and piped is None and planned is None
dist.barrier()
dist.destroy_process_group()
"""


def test_two_ranks_real_kernel_shard_and_gather(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    kids = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        kids.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                     stderr=subprocess.PIPE, text=True))
    outs = [k.communicate(timeout=600) for k in kids]
    for k, (so, se) in zip(kids, outs):
        assert k.returncode == 0, se[-3000:]
    assert "TWO_RANKS_OK" in outs[0][0]
