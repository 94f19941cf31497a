"""
Multi-GPU sampling: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI; "gloo" in CPU tests), cells sharded by BRANCH.

The reference is single-process (no collective anywhere), so this
layer is new.  The path shards naturally (SURVEY section 8 e): given the mean tensor, cells
are independent.  Every rank

  1. builds the tree together with the others (``simulate_lineage_sharded``): the host draws are the
     same on every rank (same numpy seed), a batch of candidate programs is evaluated on a SLICE OF
     THE GENES per rank with one all-reduce of the few scalars that decide acceptance, and the
     accepted branch is materialised only by the rank that owns it -- or, with the single-process
     ``simulation.simulate_lineage``, holds a replica of the whole tree,
  2. receives the same sampling plan (rank 0's plan is broadcast),
  3. owns a disjoint set of branches (greedy balance of cells per branch) and samples only
     the cells on them, keyed by their position in the GLOBAL plan (``cell_index``), so the
     count of cell n is the same whether 1, 2, 4 or 8 GPUs ran, and
  4. keeps its shard on its own device.  There is no collective on the data path; the one
     optional exchange is ``gather_rows``: count rows to rank 0 by point-to-point transfers
     (shards are unequal, so not ncclGather), all senders at once -- each drives its own xGMI
     link into the root -- in chunks, the scatter of one chunk under the transfer of the next;
     ``sample_and_gather`` is sampling and exchange as one pipeline (chunk c travels while chunk
     c + 1 is sampled; the root receives straight into final rows and returns the permutation).
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


def _comm_device(group=None):
    """Where collective buffers live: the current GPU under RCCL ("nccl"), the host under gloo."""
    import torch
    if _dist().get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _host_backend_fence(group=None):
    """RCCL orders a transfer behind the work already enqueued on the current stream, and the work enqueued later behind
    the transfer.  A HOST backend (gloo: the functional path of the tests and of PROSSTT_BENCH_BACKEND=gloo) reads and
    writes the device buffer it is handed from its own CPU threads, knowing nothing of streams: the kernels that write a
    buffer must have finished before it is posted, and the kernels that read a landed buffer before its memory is reused."""
    import torch
    if _dist().get_backend(group) != "nccl" and torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


def assign_branches_by_density(tree, world_size):
    """owner[label] -> rank before any cell exists: longest-processing-time greedy on the density mass
    of the branches (cells are drawn from it; ties by position in ``tree.branches``)."""
    mass = [float(np.sum(tree.density[b])) for b in tree.branches]
    order = sorted(range(len(tree.branches)), key=lambda i: (-mass[i], i))
    load = [0.0] * world_size
    owner = {}
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[tree.branches[i]] = r
        load[r] += mass[i]
    return owner


def simulate_lineage_sharded(tree, rel_exp_cutoff=8, intra_branch_tol=0.5, inter_branch_tol=0, *, group=None,
                             max_attempts=None, stats=None, batch=16, keep_on_device=False, **kwargs):
    """``simulation.simulate_lineage`` (simulation.py:215-286) by the ranks of ``group`` together.

    Every rank makes the same host draws (seed numpy identically); a batch of attempts is evaluated on the
    rank's slice of the genes (``lineage_attempt``: the maximum and the per-sibling counts of
    anticorrelated genes are a max and sums over genes) and ONE small all-reduce per batch -- B maxima, B x
    siblings counts -- makes every rank take the same decision; the accepted branch is committed only by
    the rank that owns it (``assign_branches_by_density``), which is also the rank that will sample its
    cells.  No rank ever holds the relative means or the mean tensor of the whole tree.

    Returns ``(rel_means, programs, coefficients)`` like the reference; ``rel_means`` holds this rank's
    branches only (``tree.resident_branches()``).  ``simulate_base_gene_exp`` and ``Tree.add_genes`` then
    work on the resident rows (the per-gene maximum over the whole tree has been all-reduced), and
    ``sample_density_sharded`` samples every cell on the rank that holds its branch."""
    import torch
    dist = _dist()
    rank, size = world(group)
    if size == 1:
        return sim.simulate_lineage(tree, rel_exp_cutoff, intra_branch_tol, inter_branch_tol,
                                    max_attempts=max_attempts, stats=stats, batch=batch, keep_on_device=keep_on_device,
                                    **kwargs)
    if not len(tree.time) == tree.num_branches:
        raise ValueError("the parameters are not enough for %i branches" % tree.num_branches)
    ctx = _device.get_context()
    dev = _comm_device(group)
    coefficients = sim.simulate_coefficients(tree, **kwargs)              # same draws on every rank
    G = tree.G
    lo, hi = rank * G // size, (rank + 1) * G // size                      # this rank's genes
    H = ctx.tensor(coefficients, torch.float64)                            # K x G: commits need every gene
    H_slice = ctx.tensor(np.ascontiguousarray(coefficients[:, lo:hi]), torch.float64) if hi > lo else None
    owner = assign_branches_by_density(tree, size)
    tree._branch_owner = owner
    tree._resident = [b for b in tree.branches if owner[b] == rank]
    offsets, rows = tree.row_offsets()
    rel = torch.empty((rows, G), dtype=torch.float64, device=ctx.torch_device)
    gene_max = torch.full((G,), -np.inf, dtype=torch.float64, device=ctx.torch_device)

    def evaluate(candidates, sibling_programs):
        B, n_sib = len(candidates), len(sibling_programs)
        if H_slice is not None:
            tops, counts = ctx.lineage_attempt_batch(candidates, H_slice, sibling_programs)
        else:                                                             # more ranks than genes
            tops, counts = np.full(B, -np.inf), np.zeros((B, n_sib), np.int64)
        t_top = torch.as_tensor(np.ascontiguousarray(tops, dtype=np.float64)).to(dev)
        dist.all_reduce(t_top, op=dist.ReduceOp.MAX, group=group)
        if n_sib:
            t_cnt = torch.as_tensor(np.ascontiguousarray(counts, dtype=np.int64)).to(dev)
            dist.all_reduce(t_cnt, op=dist.ReduceOp.SUM, group=group)
            counts = t_cnt.cpu().numpy()
        return t_top.cpu().numpy(), counts

    def commit(key, accepted):
        if owner[key] == rank:
            at = offsets[key]
            ctx.lineage_commit(accepted, H, rel[at:at + int(tree.time[key])], gene_max)

    programs = sim._lineage_loop(tree, ctx, evaluate, commit, rel_exp_cutoff, intra_branch_tol, inter_branch_tol,
                                 max_attempts, stats, "numpy", None, batch)
    # the per-gene maximum over the WHOLE tree (simulate_base_gene_exp needs it; G doubles)
    g_all = gene_max.to(dev)
    dist.all_reduce(g_all, op=dist.ReduceOp.MAX, group=group)
    gene_max = g_all.to(ctx.torch_device)
    return sim._finish_lineage(tree, rel, gene_max, H, programs, offsets, coefficients, keep_on_device)


def cells_of_rank(branch_of_cell, owner, rank):
    """Global indices (ascending) of the cells whose branch ``owner`` gives to ``rank``."""
    labels = np.asarray(branch_of_cell)
    mine = np.zeros(len(labels), dtype=bool)
    for label, r in owner.items():
        if r == rank:
            mine |= labels == label
    return np.nonzero(mine)[0].astype(np.int64)


def broadcast_plan(plan, group=None, src=0):
    """Make every rank use rank ``src``'s ``(pseudotime, branches, scalings, seed)``.

    Tensor broadcasts, not pickles: a header (N, number of distinct labels, the two halves of the
    seed), then pseudotime (int64), branch codes (int32) and scalings (float64) -- 20 B per cell; only
    the handful of distinct branch labels travels as Python objects.  ``src`` is a rank of ``group``."""
    import torch
    dist = _dist()
    rank, size = world(group)
    if size == 1:
        return plan
    dev = _comm_device(group)
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    if rank == src:
        pt, br, sc, seed = plan
        labels, codes = np.unique(np.asarray(br), return_inverse=True)
        head = torch.tensor([len(pt), len(labels), int(seed) & 0xffffffff, (int(seed) >> 32) & 0xffffffff],
                            dtype=torch.int64, device=dev)
    else:
        head = torch.zeros(4, dtype=torch.int64, device=dev)
    dist.broadcast(head, src=gsrc, group=group)
    n, n_labels, lo, hi = (int(v) for v in head.tolist())
    box = [list(labels) if rank == src else None]
    dist.broadcast_object_list(box, src=gsrc, group=group)            # a few labels (str or int), not the plan
    labels = np.asarray(box[0])
    if rank == src:
        t_pt = torch.as_tensor(np.ascontiguousarray(pt, dtype=np.int64)).to(dev)
        t_br = torch.as_tensor(np.ascontiguousarray(codes, dtype=np.int32)).to(dev)
        t_sc = torch.as_tensor(np.ascontiguousarray(sc, dtype=np.float64)).to(dev)
    else:
        t_pt = torch.empty(n, dtype=torch.int64, device=dev)
        t_br = torch.empty(n, dtype=torch.int32, device=dev)
        t_sc = torch.empty(n, dtype=torch.float64, device=dev)
    for t in (t_pt, t_br, t_sc):
        dist.broadcast(t, src=gsrc, group=group)
    if rank == src:
        return plan
    return t_pt.cpu().numpy(), labels[t_br.cpu().numpy()], t_sc.cpu().numpy(), lo | (hi << 32)


def _digest(*arrays):
    """Order-sensitive 64-bit digest of host arrays (what the ranks must agree on)."""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def assert_replicas_agree(tree, alpha, beta, group=None):
    """Every rank must hold the same tree and per-gene parameters (the lineage is replicated, not
    exchanged): compare a digest of (row sums of the device mean tensor, alpha, beta) across ranks.
    The digest (a device reduction and a small copy back) is computed once per mean tensor and cached
    on the tree; what is exchanged per call is the 16-character digest -- control path, not data path."""
    rank, size = world(group)
    if size == 1:
        return
    means = tree.device_means()
    ab = _digest(np.asarray(alpha, np.float64), np.asarray(beta, np.float64))
    key = (id(means), means.data_ptr(), tuple(means.shape))
    cached = getattr(tree, "_replica_digest", None)
    if cached is None or cached[0] != key:
        cached = (key, _digest(means.double().sum(dim=1).cpu().numpy()))
        tree._replica_digest = cached
    mine = cached[1] + ab
    seen = [None] * size
    _dist().all_gather_object(seen, mine, group=group)
    if len(set(seen)) != 1:
        raise RuntimeError("ranks hold different trees or (alpha, beta): digests %r -- seed numpy identically "
                           "on every rank before building the tree" % (seen,))


def sample_density_sharded(tree, no_cells, alpha=0.3, beta=2, scale=True, scale_v=0.7, scale_mean=0.,
                           *, seed=None, group=None, strict=True, order="presented"):
    """``simulation.sample_density`` (simulation.py:416-471) across the ranks of ``group``.

    Returns ``(counts, cell_index, sample_pt, branches, scalings)``: ``counts`` is this rank's
    int32 device tensor (len(cell_index), G); the last three are the GLOBAL plan, identical on
    every rank.  ``counts[i]`` equals row ``cell_index[i]`` of the single-GPU result for the
    same seed.

    ROW ORDER.  With ``order="presented"`` (default) the rank's cells are presented to the sampler grouped by their row of
    the mean tensor (``simulation.draw_counts`` says why), and ``cell_index`` comes back in THAT order: it is NOT
    ascending.  Label rows by ``sample_pt[cell_index]`` / ``branches[cell_index]`` -- never by a separately computed
    ascending list of the rank's cells.  ``order="plan"`` presents the rank's cells in ascending plan position
    (``cell_index`` ascending, as until round 4): 2 to 5 % slower, more HBM traffic on trees with many branches.

    EVERY rank draws the plan, the scalings and the default seed from its own numpy stream (the
    draws of the single-process call), then rank 0's values are broadcast: ranks that were seeded
    alike stay in lock-step for whatever they draw next, and ranks that were not still sample one
    plan.  With ``strict`` the ranks also compare a digest of their mean tensor, alpha and beta."""
    rank, size = world(group)
    pt, br = sim._density_plan(tree, no_cells)
    sc = sut.calc_scalings(no_cells, scale, scale_mean, scale_v)
    if seed is None:
        lo, hi = np.random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        seed = int(lo) | (int(hi) << 32)
    pt, br, sc, seed = broadcast_plan((pt, br, sc, seed), group)
    alpha = np.full(tree.G, alpha, np.float64) if np.ndim(alpha) == 0 else np.asarray(alpha, np.float64)
    beta = np.full(tree.G, beta, np.float64) if np.ndim(beta) == 0 else np.asarray(beta, np.float64)
    if tree._branch_owner is not None and size > 1:
        # the tree was built sharded: a cell is sampled where its branch's rows are
        mine = cells_of_rank(br, tree._branch_owner, rank)
    else:
        mine, _ = shard_cells(br, rank, size)
        if strict:
            assert_replicas_agree(tree, alpha, beta, group)
    ctx = _device.get_context()
    rows = sim.cell_rows(tree, pt[mine], br[mine])
    # the rank's cells are presented grouped by their row of the mean tensor (simulation.draw_counts says why); row i of
    # `counts` is the cell at position `mine[i]` of the plan, which is what every consumer of the pair goes by
    if order not in ("presented", "plan"):
        raise ValueError("order must be 'presented' or 'plan'")
    means = tree.device_means()
    mine = np.asarray(mine, dtype=np.int64)
    if order == "presented":
        perm = _device.plan_order(rows, means.shape[0])
        mine, rows = mine[perm], rows[perm]
    counts = ctx.sample_counts(means, rows, sc[mine], alpha, beta, seed=seed, cell_index=mine, check_domain=strict)
    return counts, mine, pt, br, sc


def presentation_key(tree, pseudotime, branches):
    """Row of every cell in the mean tensor of the WHOLE tree (branches in ``tree.branches`` order, time inside the
    branch): what a rank groups its cells by when it presents them to the sampler (``simulation.draw_counts`` says why).
    The same on every rank, whether or not the rank holds the branch's rows -- so every rank knows the order of every
    other rank's shard without an exchange."""
    bt = tree.branch_times()
    first, at = {}, 0
    for b in tree.branches:
        first[b] = at - bt[b][0]
        at += int(tree.time[b])
    labels = np.asarray(branches)
    key = np.asarray(pseudotime, dtype=np.int64).copy()
    for b in np.unique(labels):
        key[labels == b] += first[sim._plain_label(tree, b)]
    return key


def shards_in_presentation_order(tree, pseudotime, branches, size):
    """[cells of rank 0, cells of rank 1, ...]: positions in the plan, each shard in its order of presentation."""
    key = presentation_key(tree, pseudotime, branches)
    shards = []
    for r in range(size):
        if tree._branch_owner is not None and size > 1:
            cells = cells_of_rank(branches, tree._branch_owner, r)
        else:
            cells = shard_cells(branches, r, size)[0]
        shards.append(cells[np.argsort(key[cells], kind="stable")])
    return shards


def sample_and_gather(tree, no_cells, alpha=0.3, beta=2, scale=True, scale_v=0.7, scale_mean=0., *, seed=None,
                      group=None, dst=0, order="shard", chunk_cells=None, chunk_bytes=256 << 20, strict=True):
    """``sample_density_sharded`` and the one exchange of the path as a PIPELINE (SURVEY section 8 e): every rank samples
    its cells a chunk at a time and posts chunk c's transfer to rank ``dst`` while chunk c + 1 is being sampled -- the
    exchange (C4: 14 GB into the root at 7 x 153 GB/s >= 13 ms) is several times the sampling of a rank's share, so the
    sampling disappears under it instead of preceding it.

    order="shard" (default): the root receives every chunk STRAIGHT INTO ITS FINAL ROWS -- the matrix holds rank 0's
        cells, then rank 1's, ..., each shard in its order of presentation -- and the permutation comes back instead of a
        scatter: row i is the cell at position ``cell_of_row[i]`` of the plan.  No staging buffer, no second pass over
        the matrix on the root (``gather_rows`` lands every row in staging and copies it again).  The layout needs no
        exchange either: every rank derives every shard from the broadcast plan.
    order="plan": rows in plan order (row i = cell i), chunks staged and scattered on the root as ``gather_rows`` does,
        one round behind the transfers.

    Returns ``(counts, cell_of_row, sample_pt, branches, scalings)``; ``counts`` is the (no_cells, G) int32 device
    tensor on ``dst`` and None elsewhere; ``cell_of_row`` is None for order="plan".  Counts equal the single-process
    matrix cell for cell (they are keyed by the cell's position in the plan)."""
    import torch
    dist = _dist()
    if order not in ("shard", "plan"):
        raise ValueError("order must be 'shard' or 'plan'")
    rank, size = world(group)
    pt, br = sim._density_plan(tree, no_cells)
    sc = sut.calc_scalings(no_cells, scale, scale_mean, scale_v)
    if seed is None:
        lo32, hi32 = np.random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        seed = int(lo32) | (int(hi32) << 32)
    pt, br, sc, seed = broadcast_plan((pt, br, sc, seed), group)
    alpha = np.full(tree.G, alpha, np.float64) if np.ndim(alpha) == 0 else np.asarray(alpha, np.float64)
    beta = np.full(tree.G, beta, np.float64) if np.ndim(beta) == 0 else np.asarray(beta, np.float64)
    if strict and size > 1 and tree._branch_owner is None:
        assert_replicas_agree(tree, alpha, beta, group)
    shards = shards_in_presentation_order(tree, pt, br, size)
    mine = shards[rank]
    sizes = [len(c) for c in shards]
    first_row = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ctx = _device.get_context()
    means = tree.device_means()
    token = tree.means_token()
    rows = sim.cell_rows(tree, pt[mine], br[mine])
    G = tree.G
    dev = ctx.torch_device
    chunk = int(chunk_cells) if chunk_cells else max(1, int(chunk_bytes) // (4 * G))
    rounds = (max(sizes) + chunk - 1) // chunk if max(sizes) else 0
    is_root = rank == dst

    def peer(r):                     # P2POp addresses GLOBAL ranks
        return dist.get_global_rank(group, r) if group is not None else r

    if is_root:
        out = torch.empty((no_cells, G), dtype=torch.int32, device=dev)
        # the root's own cells: straight into their final rows (order="shard"), or a local block scattered at the end
        local = out[first_row[rank]:first_row[rank] + sizes[rank]] if order == "shard" else \
            torch.empty((sizes[rank], G), dtype=torch.int32, device=dev)
    else:
        out = None
        local = torch.empty((sizes[rank], G), dtype=torch.int32, device=dev)

    def sample(r):
        lo, hi = r * chunk, min((r + 1) * chunk, sizes[rank])
        if lo < hi:
            ctx.sample_counts(means, rows[lo:hi], sc[mine[lo:hi]], alpha, beta, seed=seed, cell_index=mine[lo:hi],
                              out=local[lo:hi], check_domain="deferred" if strict else False, means_token=token)
        return lo, hi

    def post_receives(r):
        """The root's receives of round r from every sender that still has rows: [(src, lo, hi, buffer)], requests."""
        got, ops = [], []
        for src in range(size):
            lo, hi = r * chunk, min((r + 1) * chunk, sizes[src])
            if src == dst or lo >= hi:
                continue
            buf = out[first_row[src] + lo:first_row[src] + hi] if order == "shard" else \
                torch.empty((hi - lo, G), dtype=torch.int32, device=dev)
            got.append((src, lo, hi, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer(src), group))
        return got, (dist.batch_isend_irecv(ops) if ops else [])

    def land(got):
        if order == "plan":
            for src, lo, hi, buf in got:
                out.index_copy_(0, torch.as_tensor(shards[src][lo:hi], dtype=torch.int64, device=dev), buf)
            if got:
                _host_backend_fence(group)                     # (gloo only: the staging buffers may be reused now)

    # A sender keeps at most two rounds of sends in flight (round r is posted once round r - 2 has left): the root posts
    # round r's receives only after round r - 1 has arrived, and an unbounded queue of sends has nowhere to go on a
    # backend with a fixed number of point-to-point channels.
    sends, pending = [], ([], [])
    for r in range(rounds):
        lo, hi = sample(r)                                   # enqueued; the transfers below are ordered behind it
        if size > 1 and not is_root and lo < hi:
            if len(sends) >= 2:
                for req in sends[-2]:
                    req.wait()
            _host_backend_fence(group)                         # (gloo only: the chunk has been sampled)
            sends.append(dist.batch_isend_irecv([dist.P2POp(dist.isend, local[lo:hi], peer(dst), group)]))
        if size > 1 and is_root:
            got, reqs = pending
            for req in reqs:
                req.wait()
            pending = post_receives(r)                       # round r travels while round r + 1 is sampled
            land(got)                                        # (order="plan": round r - 1 is scattered under it)
    for reqs_ in sends[-2:]:
        for req in reqs_:
            req.wait()
    if is_root:
        got, reqs = pending
        for req in reqs:
            req.wait()
        land(got)
        if order == "plan":
            out.index_copy_(0, torch.as_tensor(mine, dtype=torch.int64, device=dev), local)
    if strict:
        ctx.domain_status()
    cell_of_row = np.concatenate(shards) if order == "shard" else None
    return out, cell_of_row, pt, br, sc


def gather_rows(local_rows, cell_index, total_rows, group=None, dst=0, chunk_rows=None, chunk_bytes=256 << 20,
                to_host=False, index_of_rank=None):
    """Collect row shards on rank ``dst`` (a rank of ``group``) into a (total_rows, G) tensor in global order.

    ``local_rows`` (n_local, G) and ``cell_index`` (n_local,) of every rank; returns the full
    tensor on ``dst`` and None elsewhere.  Only rows travel in the rounds: where a row goes is known before -- from
    ``index_of_rank`` (the cell indices of EVERY rank's shard, in the order of its rows: every rank can derive them from
    the broadcast plan, ``shards_in_presentation_order``: then nothing but rows is exchanged at all), else each sender's
    index vector travels once, in front of its rows (8 B per row; until round 5 an index tensor went with every chunk, and
    the shard sizes as pickled objects).  Point-to-point (shards are unequal, so not a gather
    collective), in rounds of ``chunk_rows`` rows per sender (default: ``chunk_bytes`` = 256 MB per
    sender, so the root stages 2 rounds x (size - 1) x 256 MB whatever G is): the root posts the
    receives of a round from ALL senders at once (``batch_isend_irecv``: every xGMI link of the root
    carries data at the same time) and scatters round r with ``index_copy_`` while round r + 1 is in
    flight.  A rank may own no rows.

    ``to_host=True``: the full matrix is assembled in HOST memory on ``dst`` (page-locked when the allocator
    grants it) and returned as an ndarray -- the root then stages only the rounds in flight on its device, so a
    result larger than one GPU's memory (C5: 120 GB of int32 counts) still has somewhere to go."""
    import torch
    dist = _dist()
    rank, size = world(group)
    index = torch.as_tensor(np.asarray(cell_index), dtype=torch.int64, device=local_rows.device)
    G = local_rows.shape[1]
    def host_matrix():
        try:
            return torch.empty((total_rows, G), dtype=local_rows.dtype, pin_memory=True)
        except RuntimeError:
            return torch.empty((total_rows, G), dtype=local_rows.dtype)

    def scatter_to_host(host, idx, buf, step=1 << 16):
        """host[idx] = buf, a bounded piece at a time (the device-to-host copy of a piece, then the row scatter)."""
        idx_h = idx.cpu()
        for lo in range(0, buf.shape[0], step):
            host.index_copy_(0, idx_h[lo:lo + step], buf[lo:lo + step].cpu())

    if size == 1:
        if to_host:
            host = host_matrix()
            scatter_to_host(host, index, local_rows)
            return host.numpy()
        out = torch.empty((total_rows, G), dtype=local_rows.dtype, device=local_rows.device)
        out.index_copy_(0, index, local_rows)
        return out
    if chunk_rows is None:
        chunk_rows = max(1, int(chunk_bytes) // max(1, G * local_rows.element_size()))

    def peer(r):                     # P2POp addresses GLOBAL ranks
        return dist.get_global_rank(group, r) if group is not None else r

    comm_dev = local_rows.device
    if index_of_rank is not None:
        if len(index_of_rank) != size or len(index_of_rank[rank]) != local_rows.shape[0]:
            raise ValueError("index_of_rank must hold one index vector per rank, this rank's as long as its rows")
        sizes = [int(len(a)) for a in index_of_rank]
        where = [torch.as_tensor(np.asarray(a), dtype=torch.int64, device=comm_dev) if rank == dst else None for a in index_of_rank]
    else:
        mine_n = torch.tensor([int(local_rows.shape[0])], dtype=torch.int64, device=comm_dev)
        all_n = [torch.zeros(1, dtype=torch.int64, device=comm_dev) for _ in range(size)]
        dist.all_gather(all_n, mine_n, group=group)
        sizes = [int(t.item()) for t in all_n]
        # every sender's index vector, once
        where = [None] * size
        if rank == dst:
            ops = []
            for src in range(size):
                if src != dst and sizes[src]:
                    where[src] = torch.empty(sizes[src], dtype=torch.int64, device=comm_dev)
                    ops.append(dist.P2POp(dist.irecv, where[src], peer(src), group))
            for req in (dist.batch_isend_irecv(ops) if ops else []):
                req.wait()
        elif sizes[rank]:
            index = index.contiguous()
            _host_backend_fence(group)
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, index, peer(dst), group)]):
                req.wait()
    rounds = max((n + chunk_rows - 1) // chunk_rows for n in sizes) if max(sizes) else 0
    if rank != dst:
        rows = local_rows.contiguous()
        _host_backend_fence(group)                             # (gloo only: whatever produced the rows has finished)
        for r in range(rounds):
            lo, hi = r * chunk_rows, min((r + 1) * chunk_rows, sizes[rank])
            if lo >= hi:
                break
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, rows[lo:hi], peer(dst), group)]):
                req.wait()
        return None
    if to_host:
        out = host_matrix()
        scatter_to_host(out, index, local_rows)
    else:
        out = torch.empty((total_rows, G), dtype=local_rows.dtype, device=local_rows.device)
        out.index_copy_(0, index, local_rows)

    def post(r):
        """Receives of round r from every sender that still has rows: [(idx, buf)], requests."""
        bufs, ops = [], []
        for src in range(size):
            n = min((r + 1) * chunk_rows, sizes[src]) - r * chunk_rows
            if src == dst or n <= 0:
                continue
            buf = torch.empty((n, G), dtype=local_rows.dtype, device=local_rows.device)
            bufs.append((where[src][r * chunk_rows:r * chunk_rows + n], buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer(src), group))
        return bufs, (dist.batch_isend_irecv(ops) if ops else [])

    pending = post(0) if rounds else ([], [])
    for r in range(rounds):
        bufs, reqs = pending
        for req in reqs:
            req.wait()
        pending = post(r + 1) if r + 1 < rounds else ([], [])
        for idx, buf in bufs:
            if to_host:
                scatter_to_host(out, idx, buf)
            else:
                out.index_copy_(0, idx, buf)
        if bufs and not to_host:
            _host_backend_fence(group)                         # (gloo only: the staging buffers may be reused now)
    return out.numpy() if to_host else out
