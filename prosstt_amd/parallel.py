"""
Multi-GPU sampling: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI; "gloo" in CPU tests), cells sharded by BRANCH.

The reference is single-process (no collective anywhere), so this
layer is new.  The path shards naturally (SURVEY section 8 e): given the mean tensor, cells
are independent.  Every rank

  1. holds the same tree (the lineage stage is deterministic under the same numpy seed;
     its data -- programs, coefficients, 4*sum(T)*G bytes of means -- is small next to
     288 GB of HBM, so it is replicated rather than exchanged),
  2. receives the same sampling plan (rank 0's plan is broadcast),
  3. owns a disjoint set of branches (greedy balance of cells per branch) and samples only
     the cells on them, keyed by their position in the GLOBAL plan (``cell_index``), so the
     count of cell n is the same whether 1, 2, 4 or 8 GPUs ran, and
  4. keeps its shard on its own device.  There is no collective on the data path; the one
     optional exchange is ``gather_rows``: count rows to rank 0 by point-to-point send/recv
     (shards are unequal, so not ncclGather), each sender driving one xGMI link.
"""
import numpy as np

from . import device as _device
from . import simulation as sim
from . import sim_utils as sut


def _dist():
    import torch.distributed as dist
    return dist


def world(group=None):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def assign_branches_to_ranks(branch_of_cell, world_size):
    """owner[label] -> rank.  Longest-processing-time greedy on cells per branch: branches
    in decreasing cell count (ties by first appearance) go to the least-loaded rank."""
    labels, first, counts = np.unique(np.asarray(branch_of_cell), return_index=True, return_counts=True)
    order = sorted(range(len(labels)), key=lambda i: (-counts[i], first[i]))
    load = [0] * world_size
    owner = {}
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[labels[i].item() if hasattr(labels[i], "item") else labels[i]] = r
        load[r] += int(counts[i])
    return owner


def shard_cells(branch_of_cell, rank, world_size):
    """Global indices (ascending) of the cells that ``rank`` owns, and the branch owner map."""
    owner = assign_branches_to_ranks(branch_of_cell, world_size)
    labels = np.asarray(branch_of_cell)
    owner_of_cell = np.empty(len(labels), dtype=np.int64)
    for label, r in owner.items():
        owner_of_cell[labels == label] = r
    return np.nonzero(owner_of_cell == rank)[0].astype(np.int64), owner


def broadcast_plan(plan, group=None, src=0):
    """Make every rank use rank ``src``'s (pseudotime, branches, scalings)."""
    rank, size = world(group)
    if size == 1:
        return plan
    box = [plan if rank == src else None]
    _dist().broadcast_object_list(box, src=src, group=group)
    return box[0]


def sample_density_sharded(tree, no_cells, alpha=0.3, beta=2, scale=True, scale_v=0.7, scale_mean=0.,
                           *, seed=None, group=None, strict=True):
    """``simulation.sample_density`` (simulation.py:416-471) across the ranks of ``group``.

    Returns ``(counts, cell_index, sample_pt, branches, scalings)``: ``counts`` is this rank's
    int32 device tensor (len(cell_index), G); the last three are the GLOBAL plan, identical on
    every rank.  ``counts[i]`` equals row ``cell_index[i]`` of the single-GPU result for the
    same seed."""
    rank, size = world(group)
    if rank == 0:
        pt, br = sim._density_plan(tree, no_cells)
        sc = sut.calc_scalings(no_cells, scale, scale_mean, scale_v)
        if seed is None:
            lo, hi = np.random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
            seed = int(lo) | (int(hi) << 32)
        plan = (pt, br, sc, seed)
    else:
        plan = None
    pt, br, sc, seed = broadcast_plan(plan, group)
    mine, _ = shard_cells(br, rank, size)
    if np.shape(alpha) == ():
        alpha = [alpha] * tree.G
    if np.shape(beta) == ():
        beta = [beta] * tree.G
    ctx = _device.get_context()
    rows = sim.cell_rows(tree, pt[mine], br[mine])
    counts = ctx.sample_counts(tree.device_means(), rows, sc[mine], np.asarray(alpha, dtype=np.float64),
                               np.asarray(beta, dtype=np.float64), seed=seed, cell_index=mine,
                               check_domain=strict)
    return counts, mine, pt, br, sc


def gather_rows(local_rows, cell_index, total_rows, group=None, dst=0):
    """Collect row shards on rank ``dst`` into a (total_rows, G) tensor in global order.

    ``local_rows`` (n_local, G) and ``cell_index`` (n_local,) of every rank; returns the full
    tensor on ``dst`` and None elsewhere.  Point-to-point: every sender streams its shard over
    its own link, the root receives into staging and scatters rows with ``index_copy_``."""
    import torch
    dist = _dist()
    rank, size = world(group)
    index = torch.as_tensor(np.asarray(cell_index), dtype=torch.int64, device=local_rows.device)
    if size == 1:
        out = torch.empty((total_rows, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        out.index_copy_(0, index, local_rows)
        return out
    sizes = [None] * size
    dist.all_gather_object(sizes, int(local_rows.shape[0]), group=group)
    if rank == dst:
        out = torch.empty((total_rows, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
        out.index_copy_(0, index, local_rows)
        for src in range(size):
            if src == dst or sizes[src] == 0:
                continue
            idx = torch.empty(sizes[src], dtype=torch.int64, device=local_rows.device)
            buf = torch.empty((sizes[src], local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
            dist.recv(idx, src=src, group=group)
            dist.recv(buf, src=src, group=group)
            out.index_copy_(0, idx, buf)
        return out
    if local_rows.shape[0]:
        dist.send(index, dst=dst, group=group)
        dist.send(local_rows.contiguous(), dst=dst, group=group)
    return None
