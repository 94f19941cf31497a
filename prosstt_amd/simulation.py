"""
Simulation entry points, mirroring ``prosstt.simulation``
(reference: /root/reference/prosstt/simulation.py).  Same names, positional
arguments, defaults and return values; new options are keyword-only.

Division of labour
 * host (numpy, same RandomState calls in the same order as the reference):
   random-walk variates, coefficient draws, the (pseudotime, branch) plan of the
   cells, library-size factors -- O(branches*K*T), O(K*G) or O(cells) numbers;
 * device (hand-written HIP, through the C ABI of include/prosstt_amd.h):
   ``programs @ coefficients`` with the accept/reject reductions, the mean tensor,
   and the fused negative-binomial count sampler -- O(T*G*K) and O(cells*genes).

There is no CPU fallback for the device part.
"""
import warnings

import numpy as np
from numpy import random
import pandas as pd

from . import count_model as cm
from . import device as _device
from . import sim_utils as sut
from .device import HOST_OUTS as _HOST_OUTS, OUT_CHOICES as _OUT_CHOICES, host_return as _host_return


# ----------------------------------------------------------------------------------
# expression programs and coefficients (host draws)
# ----------------------------------------------------------------------------------

def _walk_loop(log_start, vel0, eta, eps):
    """The reference's recurrence, step by step (simulation.py:114-121)."""
    steps = len(eps) + 1
    walk = np.zeros(steps)
    velocity = np.zeros(steps)
    walk[0] = log_start
    velocity[0] = vel0
    for t in range(steps - 1):
        walk[t + 1] = walk[t] + velocity[t]
        velocity[t + 1] = eta * velocity[t] + eps[t]
    return walk


def _walk_filter(log_start, vel0, eta, eps):
    """Same binary64 operations in the same order, without the Python loop: the velocity
    is the IIR filter y[n] = x[n] + eta*y[n-1] and the walk a left-to-right cumulative sum."""
    from scipy.signal import lfilter
    velocity = lfilter([1.0], [1.0, -eta], np.concatenate(([vel0], eps)))
    return np.cumsum(np.concatenate(([log_start], velocity[:-1])))


_WALK = None


def _walk(log_start, vel0, eta, eps):
    """Pick the filter form once, after checking on this machine that it reproduces the
    loop bit for bit (it does wherever scipy's lfilter is built without fused multiply-add)."""
    global _WALK
    if _WALK is None:
        probe = np.sin(np.arange(1, 64)) * 0.04
        same = np.array_equal(_walk_loop(-0.3, 0.11, 0.73, probe), _walk_filter(-0.3, 0.11, 0.73, probe))
        _WALK = _walk_filter if same else _walk_loop
    return _WALK(log_start, vel0, eta, eps)


def diffusion(steps):
    """Random walk with momentum (simulation.py:89-124).  Variates are drawn in the
    reference's order -- U, N, U, then (steps-1) x N -- so the walk is the same
    sequence of binary64 operations on the same numbers."""
    start = random.random_sample() * 1.5 + 0
    vel0 = random.standard_normal() * 0.2 + 0
    s_eps = 2 / steps
    eta = random.random_sample() * 1 + 0
    eps = random.standard_normal(steps - 1) * s_eps + 0 if steps > 1 else np.zeros(0)
    return _walk(np.log(start), vel0, eta, eps)


def sim_expr_branch(branch_length, expr_progr, cutoff=0.2, max_loops=100):
    """(branch_length, expr_progr) matrix of independent walks (simulation.py:21-86).
    The reference's correlation filter never fires (see sim_utils.test_correlation),
    so ``cutoff`` and ``max_loops`` are accepted and ignored; a single program, which
    hangs the reference, is refused."""
    if expr_progr < 2:
        raise ValueError("at least 2 expression programs are needed (the reference never "
                         "terminates for 1)")
    programs = np.zeros((expr_progr, branch_length))
    for k in range(expr_progr):
        programs[k] = diffusion(branch_length)
    return np.transpose(programs)


def sim_expr_branches(attempts, branch_length, expr_progr):
    """``attempts`` consecutive ``sim_expr_branch(branch_length, expr_progr)`` results in one go:
    ``(programs (attempts, branch_length, expr_progr), states)`` where ``states[i]`` is what
    ``np.random.get_state()`` would return behind the i-th call.  The variates come from numpy's own global
    stream through the library's generator of it (``prosstt_amd_numpy_programs``: one call instead of
    4 * expr_progr * attempts numpy calls), the walks are the reference's recurrence (simulation.py:114-121)
    advanced for all of them at once, one time step per numpy operation: the same binary64 operations on the
    same numbers.  numpy's generator is left behind the LAST attempt; the caller rewinds it with
    ``np.random.set_state(states[i])``."""
    import ctypes
    from . import _native
    if expr_progr < 2:
        raise ValueError("at least 2 expression programs are needed (the reference never "
                         "terminates for 1)")
    lib = _native.load()
    kind, words, pos, has_gauss, gauss = random.get_state()
    if kind != "MT19937":
        raise RuntimeError("numpy's global generator is not the legacy MT19937")
    words = np.ascontiguousarray(words, dtype=np.uint32).copy()
    pos_c, has_c, gauss_c = ctypes.c_int32(int(pos)), ctypes.c_int32(int(has_gauss)), ctypes.c_double(float(gauss))
    B, T, K = int(attempts), int(branch_length), int(expr_progr)
    start, vel0, eta = (np.empty((B, K)) for _ in range(3))
    noise = np.empty((B, K, max(T - 1, 0)))
    after_words = np.empty((B, 624), np.uint32)
    after_pos, after_has = np.empty(B, np.int32), np.empty(B, np.int32)
    after_gauss = np.empty(B)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _native.check(lib.prosstt_amd_numpy_programs(
        ptr(words), ctypes.byref(pos_c), ctypes.byref(has_c), ctypes.byref(gauss_c), B, T, K, ptr(start), ptr(vel0),
        ptr(eta), ptr(noise), ptr(after_words), ptr(after_pos), ptr(after_has), ptr(after_gauss)))
    random.set_state(("MT19937", words, pos_c.value, has_c.value, gauss_c.value))
    states = [("MT19937", after_words[i], int(after_pos[i]), int(after_has[i]), float(after_gauss[i])) for i in range(B)]
    walks = np.empty((B, T, K))
    level, velocity = np.log(start), vel0
    for t in range(T - 1):
        walks[:, t] = level
        level = level + velocity
        velocity = eta * velocity + noise[:, :, t]
    walks[:, T - 1] = level
    return walks, states


def simulate_coefficients(tree, fallback_a=0.04, **kwargs):
    """(K, G) contribution of every program to every gene (simulation.py:127-161)."""
    if "a" not in kwargs.keys():
        warnings.warn("No argument 'a' specified in kwargs: using gamma and a=0.04", UserWarning)
        return _sim_coeff_gamma(tree, fallback_a)
    if "b" in kwargs.keys():
        # the reference ignores the passed values and uses Beta(2, 2) (simulation.py:157-159, 164)
        return _sim_coeff_beta(tree, sut.create_groups(tree.modules, tree.G))
    return _sim_coeff_gamma(tree, a=kwargs['a'])


def _sim_coeff_beta(tree, groups, a=2, b=2):
    """simulation.py:164-189."""
    H = np.zeros((tree.modules, tree.G))
    for k in range(tree.modules):
        for gene in groups[k]:
            H[k][gene] += random.beta(a, b) * 1 + 0
    return H


def _sim_coeff_gamma(tree, a=0.05):
    """simulation.py:192-212."""
    K, G = tree.modules, tree.G
    return np.reshape(random.standard_gamma(a, K * G) * 1 + 0, (K, G))


# ----------------------------------------------------------------------------------
# lineage (device: K2 kernels)
# ----------------------------------------------------------------------------------

def _stack_rows(tree, per_branch):
    return np.concatenate([np.asarray(per_branch[b], dtype=np.float64) for b in tree.resident_branches()], axis=0)


def _lineage_cache(tree, relative_means):
    """The device tensors simulate_lineage left behind, if ``relative_means`` are the very arrays it
    returned AND their contents are untouched (fingerprint); else None -- the caller uploads."""
    cache = tree._lineage
    held = tree.resident_branches()
    if cache is None or not all(b in relative_means and relative_means[b] is cache["host"][b] for b in held):
        return None
    if _device.host_fingerprint([cache["host"][b] for b in held]) != cache["print"]:
        return None
    return cache


def _device_rel(tree, relative_means):
    """(sum T_b, G) binary64 device tensor of ``relative_means``; the tensor that simulate_lineage
    left on the device is reused when these are the unmodified arrays it returned."""
    import torch
    cache = _lineage_cache(tree, relative_means)
    if cache is not None:
        return cache["rel"]
    ctx = _device.get_context()
    return ctx.tensor(_stack_rows(tree, relative_means), torch.float64)


def _device_gene_max(tree, relative_means):
    """(G,) device tensor: max over branches and time of the relative means (log of
    sim_utils.max_relat_exp reduced over branches, sim_utils.py:460-461)."""
    import torch
    cache = _lineage_cache(tree, relative_means)
    if cache is not None:
        return cache["gene_max"]
    if getattr(tree, "_branch_owner", None) is not None:
        # built sharded over ranks: this process holds only its own branches' rows, so a maximum recomputed here
        # would differ from rank to rank and the ranks would build mutually inconsistent mean tensors
        raise ValueError("the relative means of a tree built by simulate_lineage_sharded must be passed on unmodified "
                         "(the per-gene maximum over the whole tree was reduced across the ranks when it was built)")
    ctx = _device.get_context()
    rel = ctx.tensor(_stack_rows(tree, relative_means), torch.float64)
    gmax = torch.full((rel.shape[1],), -np.inf, dtype=torch.float64, device=ctx.torch_device)
    return ctx.gene_max(rel, gmax)


def simulate_lineage(tree, rel_exp_cutoff=8, intra_branch_tol=0.5, inter_branch_tol=0,
                     *, max_attempts=None, stats=None, rng="numpy", seed=None, batch=16, keep_on_device=False, **kwargs):
    """Relative mean expression of every gene at every point of the tree
    (simulation.py:215-286).

    Returns ``(pd.Series rel_means, pd.Series programs, coefficients)`` like the
    reference.  Per branch, in breadth-first order, programs are redrawn until
    ``max(programs @ H) <= rel_exp_cutoff`` and more than ``inter_branch_tol`` of
    the genes are anticorrelated with every already-simulated sibling.  ``batch``
    attempts are evaluated by one ``lineage_attempt`` launch (no (T, G) matrix is formed,
    a few scalars come back per attempt); the first acceptable one is taken and numpy's
    stream is rewound to just behind its draws, so results and stream position equal the
    one-at-a-time loop's.  The accepted branch is materialised once by ``lineage_commit``.

    New keyword-only options: ``batch`` (attempts per launch); ``max_attempts`` bounds the redraws per branch
    (default unlimited, like the reference); ``stats`` (a list) receives one
    ``(branch, max, [anticorrelated counts])`` record per attempt; ``rng="device"``
    draws the random walks on the device (``lineage_walk`` kernel, one lane per program,
    Philox streams keyed by ``seed``, branch and attempt) instead of numpy's global stream --
    same walk law, different numbers (``seed`` defaults to two draws of numpy's stream);
    ``keep_on_device=True`` returns handles of the relative means (``DeviceRows``) instead of copying the
    (sum T_b, G) binary64 matrix to the host -- 3 GB at 256 branches x 30 000 genes -- which
    ``simulate_base_gene_exp`` and ``Tree.add_genes`` accept as they accept the arrays.
    """
    import torch
    if not len(tree.time) == tree.num_branches:
        raise ValueError("the parameters are not enough for %i branches" % tree.num_branches)
    ctx = _device.get_context()
    tree._resident = tree._branch_owner = None       # this process builds and keeps the whole tree
    coefficients = simulate_coefficients(tree, **kwargs)
    H = ctx.tensor(coefficients, torch.float64)
    offsets, rows = tree.row_offsets()
    rel = torch.empty((rows, tree.G), dtype=torch.float64, device=ctx.torch_device)
    gene_max = torch.full((tree.G,), -np.inf, dtype=torch.float64, device=ctx.torch_device)

    def evaluate(candidates, sibling_programs):
        return ctx.lineage_attempt_batch(candidates, H, sibling_programs)

    def commit(key, accepted):
        at = offsets[key]
        ctx.lineage_commit(accepted, H, rel[at:at + int(tree.time[key])], gene_max)

    programs = _lineage_loop(tree, ctx, evaluate, commit, rel_exp_cutoff, intra_branch_tol, inter_branch_tol,
                             max_attempts, stats, rng, seed, batch)
    return _finish_lineage(tree, rel, gene_max, H, programs, offsets, coefficients, keep_on_device)


def _lineage_loop(tree, ctx, evaluate, commit, rel_exp_cutoff, intra_branch_tol, inter_branch_tol,
                  max_attempts, stats, rng, seed, batch):
    """The accept/reject loop of simulate_lineage (simulation.py:264-282) over the branches in breadth-first
    order.  ``evaluate(candidates (B, T, K), sibling programs) -> (max per attempt, anticorrelated-gene counts per
    attempt and sibling)`` and ``commit(branch, programs)`` are the device side: one process evaluates whole
    genes, ``parallel.simulate_lineage_sharded`` a gene slice per rank with an all-reduce of the few scalars."""
    if batch < 1:
        raise ValueError("batch must be at least 1")
    if max_attempts is not None and max_attempts < 1:
        raise ValueError("max_attempts must be at least 1")
    if rng not in ("numpy", "device"):
        raise ValueError("rng must be 'numpy' or 'device'")
    if rng == "device" and seed is None:
        lo, hi = random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        seed = int(lo) | (int(hi) << 32)
    if rng == "device" and tree.modules < 2:
        raise ValueError("at least 2 expression programs are needed")
    topology = np.array(tree.topology)
    programs = {}
    for ordinal, branch in enumerate(sut.breadth_first_branches(tree)):
        steps, tries, accepted = int(tree.time[branch]), 0, None
        siblings = None
        while accepted is None:
            # Candidate programs for the next `width` attempts, drawn in the reference's stream order
            # (one sim_expr_branch per attempt); all of them go through ONE lineage_attempt launch.  The
            # first acceptable one is taken and numpy's generator is put back to where the sequential
            # loop would have left it, just behind that attempt's draws -- same numbers, same stream
            # position, one device round trip per batch instead of per attempt.  (Device-mode walks do not
            # touch numpy's stream, and the first attempt is usually accepted: one walk first, batches after.)
            first = 1 if rng == "device" else 4
            width = min(batch, first if tries == 0 else batch)           # most branches are accepted early
            if max_attempts is not None:
                width = min(width, max_attempts - tries)
            candidates, states = [], []
            entry_state = random.get_state() if rng == "numpy" else None
            try:
                if rng == "device":
                    drawn = ctx.lineage_walks(seed, [(ordinal << 32) | (tries + i) for i in range(width)], steps, int(tree.modules))
                else:
                    drawn, states = sim_expr_branches(width, steps, int(tree.modules))
                for i in range(width):
                    programs[branch] = drawn[i]
                    candidates.append(sut.adjust_to_parent(programs, branch, topology))
                if siblings is None:
                    siblings = [b for b in sut.find_parallel(tree, programs, branch)
                                if b is not None and b != branch]
                tops, counts = evaluate(np.stack(candidates), [programs[s] for s in siblings])
            except BaseException:
                # the stream goes back to where the batch found it: a failed call must not leave numpy
                # advanced past the position of the sequential loop
                if entry_state is not None:
                    random.set_state(entry_state)
                programs.pop(branch, None)
                raise
            for i in range(width):
                tries += 1
                if stats is not None:
                    stats.append((branch, float(tops[i]), [int(c) for c in counts[i]]))
                diverges = all(c / (tree.G * 1.0) > inter_branch_tol for c in counts[i])
                if not (tops[i] > rel_exp_cutoff) and diverges:
                    accepted = candidates[i]
                    if states:
                        random.set_state(states[i])
                    break
            if accepted is None and max_attempts is not None and tries >= max_attempts:
                del programs[branch]
                raise RuntimeError("branch %r: no acceptable expression programs after %d attempts "
                                   "(rel_exp_cutoff=%r, inter_branch_tol=%r)"
                                   % (branch, tries, rel_exp_cutoff, inter_branch_tol))
        programs[branch] = accepted
        commit(_plain_label(tree, branch), accepted)
    return programs


class DeviceRows:
    """The relative means of one branch, left on the device (``simulate_lineage(..., keep_on_device=True)``):
    ``simulate_base_gene_exp`` and ``Tree.add_genes`` take it as they take the host array; ``np.asarray`` of it
    (or indexing) copies the branch's (T_b, G) rows to the host on demand."""

    def __init__(self, rel, first_row, steps):
        self._rel, self._at, self._steps = rel, int(first_row), int(steps)
        self.shape = (int(steps), int(rel.shape[1]))
        self.dtype = np.dtype(np.float64)
        self.ndim = 2

    def __array__(self, dtype=None, copy=None):
        host = self._rel[self._at:self._at + self._steps].cpu().numpy()
        return host if dtype is None else host.astype(dtype, copy=False)

    def __getitem__(self, index):
        return np.asarray(self)[index]

    def __len__(self):
        return self._steps


def _finish_lineage(tree, rel, gene_max, H, programs, offsets, coefficients, keep_on_device=False):
    """Host copies of the resident branches' relative means (or handles of their device rows), the device cache,
    the reference's return value."""
    held = tree.resident_branches()
    if keep_on_device:
        rel_means = {b: DeviceRows(rel, offsets[b], tree.time[b]) for b in held}
    else:
        host = rel.cpu().numpy()
        rel_means = {b: host[offsets[b]:offsets[b] + int(tree.time[b])] for b in held}
    # the device keeps `rel` for simulate_base_gene_exp / add_genes, which recognise these arrays by
    # identity and fingerprint: writable like the reference's; edited arrays are uploaded afresh
    tree._lineage = dict(rel=rel, gene_max=gene_max, host=rel_means, H=H,
                         print=_device.host_fingerprint([rel_means[b] for b in held]))
    ordered = {}
    for branch in programs:                      # keep the reference's insertion (BFS) order
        key = _plain_label(tree, branch)
        if key in rel_means:
            ordered[branch] = rel_means[key]
    return pd.Series(ordered), pd.Series(programs), coefficients


def _plain_label(tree, label):
    """numpy scalar label -> the key object used in ``tree.branches``."""
    for b in tree.branches:
        if b == label:
            return b
    raise KeyError(label)


# ----------------------------------------------------------------------------------
# sampling (host plan + device K3)
# ----------------------------------------------------------------------------------

def sample_whole_tree_restricted(tree, alpha=0.2, beta=3, **device_opts):
    """Default one-shot simulation (simulation.py:289-316); returns 4 values like the
    reference does (its docstring lists 3)."""
    # one cell per pseudotime step; the stream order is: lineage, base expression, alpha/beta, branches, scalings
    tree.default_gene_expression()
    per_gene = cm.generate_negbin_params(tree, mean_alpha=alpha, mean_beta=beta)
    return _sample_data_at_times(tree, np.arange(tree.get_max_time()), alpha=per_gene[0], beta=per_gene[1],
                                 **device_opts)


def sample_pseudotime_series(tree, cells, series_points, point_std, alpha=0.3, beta=2, scale=True,
                             scale_mean=0, scale_v=0.7, **device_opts):
    """Cells normally distributed around sample time points (simulation.py:319-379)."""
    series_points, cells, point_std = sut.process_timeseries_input(series_points, cells, point_std)
    pseudotimes = []
    max_time = tree.get_max_time()
    for t, n, var in zip(series_points, cells, point_std):
        pseudotimes.extend(draw_times(t, n, max_time, var))
    return _sample_data_at_times(tree, np.array(pseudotimes), alpha=alpha, beta=beta, scale=scale,
                                 scale_mean=scale_mean, scale_v=scale_v, **device_opts)


def draw_times(timepoint, no_cells, max_time, var=4):
    """simulation.py:382-413."""
    sample_pt = (random.standard_normal(no_cells) * var + timepoint).astype(int)
    sample_pt[sample_pt < 0] = 0
    sample_pt[sample_pt >= max_time] = max_time - 1
    return sample_pt


def _density_plan(tree, no_cells):
    """(pseudotime, branch) of ``no_cells`` cells drawn from ``tree.density``: one
    ``np.random.choice`` over all sum(T_b) positions of the tree, weighted by the density
    (simulation.py:452-467 -- the same single draw, so the same cells at equal numpy seed)."""
    spans = tree.branch_times()
    weights = np.concatenate([tree.density[b] for b in tree.branches])
    picked = random.choice(np.arange(weights.size), size=no_cells, p=weights)
    # position i of the concatenation belongs to branch owner[i] at pseudotime first[owner] + offset
    lengths = np.array([tree.time[b] for b in tree.branches], dtype=np.int64)
    owner = np.repeat(np.arange(len(tree.branches)), lengths)[picked]
    first = np.array([spans[b][0] for b in tree.branches], dtype=np.int64)
    offset = picked - (np.cumsum(lengths) - lengths)[owner]
    return first[owner] + offset, np.asarray(tree.branches)[owner]


def sample_density(tree, no_cells, alpha=0.3, beta=2, scale=True, scale_v=0.7, scale_mean=0.,
                   **device_opts):
    """Sample cells according to the density along the tree (simulation.py:416-471)."""
    sample_time, sample_branches = _density_plan(tree, no_cells)
    return _sample_data_at_times(tree, sample_time, alpha=alpha, beta=beta, branches=sample_branches,
                                 scale=scale, scale_mean=scale_mean, scale_v=scale_v, **device_opts)


def sample_density_chunks(tree, no_cells, chunk_cells, alpha=0.3, beta=2, scale=True, scale_v=0.7, scale_mean=0.,
                          *, seed=None, out="numpy32", strict=True, order="presented"):
    """``sample_density`` (simulation.py:416-471) for matrices that should not exist whole on the host -- or
    not all at once: a generator of ``(counts, pseudotime, branches, scalings)`` for successive ranges of
    ``chunk_cells`` cells of ONE plan (drawn up front, with the reference's numpy calls).  Chunk i + 1 is
    sampled on the device while chunk i travels to the host; every count equals the one the single call
    returns for that cell (the sampler is keyed by the cell's position in the plan), so the chunks
    concatenate to ``sample_density``'s matrix.  ``out``, ``order``: as in ``draw_counts`` ("torch" yields
    ``device.PresentedCounts`` -- the chunk's device tensor in the order its cells were presented, and the permutation --
    or, with ``order="plan"``, plain device tensors in plan order; the consumer must be done with a chunk before asking
    for the next chunk but one)."""
    if out != "torch" and out not in _HOST_OUTS:
        raise ValueError(_OUT_CHOICES)
    if order not in ("presented", "plan"):
        raise ValueError("order must be 'presented' or 'plan'")
    if chunk_cells <= 0:
        raise ValueError("chunk_cells must be positive")
    alpha, beta = (np.full(tree.G, v, dtype=np.float64) if np.ndim(v) == 0 else np.asarray(v, dtype=np.float64)
                   for v in (alpha, beta))
    sample_time, sample_branches = _density_plan(tree, no_cells)
    scalings = sut.calc_scalings(no_cells, scale, scale_mean, scale_v)
    if seed is None:
        lo, hi = random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        seed = int(lo) | (int(hi) << 32)
    ctx = _device.get_context()
    rows = cell_rows(tree, sample_time, sample_branches)
    token = tree.means_token()
    means = tree.device_means()

    def launch(lo):
        # (presented grouped by mean-tensor row, put back in plan order inside the copy to the host: see draw_counts)
        hi = min(lo + chunk_cells, no_cells)
        if out == "torch" and order == "plan":
            return ctx.sample_counts(means, rows[lo:hi], scalings[lo:hi], alpha, beta, seed=seed, cell_offset=lo,
                                     check_domain="deferred" if strict else False, means_token=token), None
        perm = _device.plan_order(rows[lo:hi], means.shape[0])
        return ctx.sample_counts(means, rows[lo:hi][perm], scalings[lo:hi][perm], alpha, beta, seed=seed,
                                 cell_index=lo + perm.astype(np.int64),
                                 check_domain="deferred" if strict else False, means_token=token), perm

    # The verdict of the deferred domain check is sticky in the ctx and covers every chunk enqueued so far (an invalid
    # chunk i + 1 may already be reported with chunk i: the plan is one call).  Whatever ends the generator -- exhaustion,
    # an exception of the host copy, the consumer dropping it -- leaves no verdict behind for an unrelated later call.
    # (verdict_open: a checked launch has been enqueued whose verdict nobody has read yet)
    verdict_open = False
    pending = launch(0) if no_cells else None
    verdict_open = strict and pending is not None
    try:
        for lo in range(0, no_cells, chunk_cells):
            hi = min(lo + chunk_cells, no_cells)
            counts, perm = pending
            pending = launch(hi) if hi < no_cells else None      # enqueued behind `counts`, runs under its copy
            if out == "torch":
                host = counts if perm is None else _device.PresentedCounts(counts, perm)
            else:
                host = _host_return(counts, out, row_order=perm)
            if strict:
                ctx.domain_status()
                verdict_open = pending is not None               # (the next chunk's launch is already behind it)
            yield host, sample_time[lo:hi], sample_branches[lo:hi], scalings[lo:hi]
    finally:
        if verdict_open:
            _discard_verdict(ctx)


def sample_whole_tree(tree, n_factor, alpha=0.3, beta=2, scale=True, scale_mean=0., scale_v=0.7,
                      **device_opts):
    """Every (pseudotime, branch) position ``n_factor`` times (simulation.py:474-517)."""
    pseudotime, branches = cover_whole_tree(tree)
    return _sample_data_at_times(tree, np.repeat(pseudotime, n_factor), alpha=alpha, beta=beta,
                                 branches=np.repeat(branches, n_factor), scale=scale,
                                 scale_mean=scale_mean, scale_v=scale_v, **device_opts)


def cover_whole_tree(tree):
    """All (pseudotime, branch) pairs of the tree, timezone by timezone (simulation.py:520-548)."""
    timezone = tree.populate_timezone()
    assignments = sut.assign_branches(tree.branch_times(), timezone)
    pseudotime, branches = [], []
    for i, (start, last) in enumerate(timezone):
        for branch in assignments[i]:
            pseudotime.extend(np.arange(start, last + 1))
            branches.extend([branch] * (last + 1 - start))
    return pseudotime, branches


def _sample_data_at_times(tree, sample_pt, branches=None, alpha=0.3, beta=2, scale=True,
                          scale_mean=0., scale_v=0.7, **device_opts):
    """Counts for cells at given pseudotimes (simulation.py:551-599)."""
    # scalar hyper-parameters apply to every gene; branches are drawn before the scalings (stream order)
    alpha, beta = (np.full(tree.G, v) if np.ndim(v) == 0 else v for v in (alpha, beta))
    if branches is None:
        branches = sut.pick_branches(tree, sample_pt)
    scalings = sut.calc_scalings(len(sample_pt), scale, scale_mean, scale_v)
    counts = draw_counts(tree, sample_pt, branches, scalings, alpha, beta, **device_opts)
    return counts, sample_pt, branches, scalings


def cell_rows(tree, pseudotime, branches):
    """Row of every cell in the device mean tensor: row offset of its branch plus
    ``pseudotime - branch start`` (simulation.py:634-635), without the reference's
    per-cell ``branch_times()`` call."""
    bt = tree.branch_times()
    offsets, _ = tree.row_offsets()
    codes, uniques = pd.factorize(np.asarray(branches), sort=False)
    start = np.empty(len(uniques), dtype=np.int64)
    base = np.empty(len(uniques), dtype=np.int64)
    length = np.empty(len(uniques), dtype=np.int64)
    for i, label in enumerate(uniques):
        key = _plain_label(tree, label)
        if key not in offsets:
            raise ValueError("branch %r is not resident on this process (the tree is sharded: "
                             "sample the cells of a branch on the rank that owns it)" % (key,))
        start[i], base[i], length[i] = bt[key][0], offsets[key], int(tree.time[key])
    inside = np.asarray(pseudotime, dtype=np.int64) - start[codes]
    if len(inside) and (inside.min() < 0 or np.any(inside >= length[codes])):
        raise IndexError("a pseudotime value lies outside its branch")
    return (base[codes] + inside).astype(np.int32)


def draw_counts(tree, pseudotime, branches, scalings, alpha, beta, *, seed=None, out="numpy",
                strict=True, order="presented"):
    """UMI counts of every cell and gene (simulation.py:602-651).

    One launch of the fused HIP sampler: gather the cell's row of the mean tensor,
    scale it, form the negative-binomial parameters of ``count_model.get_pr_umi`` and
    draw.  Counts follow the reference's law, NB(n = r, p = 1 - p); the random
    stream is the counter-based PRNB-7, not numpy's MT19937,
    so individual values differ from the reference at equal numpy seed.

    New keyword-only options
      seed    64-bit sampler seed.  Default: two 32-bit draws from numpy's global
              stream, so ``np.random.seed`` still determines the whole simulation.
      out     "numpy" (default): int64 ndarray like the reference; "numpy32": int32 (what the
              device holds -- half the bytes over PCIe); "numpy16": uint16 (a quarter; OverflowError
              if a count does not fit); "csr": ``scipy.sparse.csr_matrix`` of int32, compacted on the device -- 8 bytes
              per NON-ZERO cross PCIe and the dense matrix never exists on the host (two thirds of a count matrix are
              zeros: ``device.to_host_csr``); "torch": no host copy -- the int32 device matrix as
              ``device.PresentedCounts`` (see ``order``).
      order   (with out="torch"; host arrays always come back in plan order) "presented" (default): the matrix as it
              lies on the device -- the cells are presented to the sampler grouped by their row of the mean tensor, which
              keeps a gene tile's rows of the tensor in cache -- with the permutation: ``counts, cell_of_row = result``,
              row i belongs to cell ``cell_of_row[i]``; ``result.in_plan_order()`` gathers.  "plan": a plain device
              tensor whose row n is cell n; the cells are then presented as planned (2 to 5 % slower, and on a
              32-branch tree 1.75 x the HBM traffic: the mean tensor's rows are fetched once per cell).
      strict  raise ``ValueError`` where scipy's argument check would (an exact-zero
              mean, or alpha*m + beta < 1); the test rides in the call's own kernels (the per-row flags of
              the mean tensor are kept until the tensor changes) and costs no launch and no extra
              synchronisation.  False skips it.
    """
    no_cells = len(branches)
    if len(pseudotime) != no_cells or len(scalings) != no_cells:
        raise ValueError("pseudotime, branches and scalings must have one entry per cell")
    if seed is None:
        lo, hi = random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        seed = int(lo) | (int(hi) << 32)
    ctx = _device.get_context()
    rows = cell_rows(tree, pseudotime, branches)
    if out != "torch" and out not in _HOST_OUTS:
        raise ValueError(_OUT_CHOICES)
    # the domain check rides in the call's own kernels and is not waited for; its verdict is read behind the copy to the
    # host (which synchronises anyway), or at once when the device tensor itself is returned
    if order not in ("presented", "plan"):
        raise ValueError("order must be 'presented' or 'plan'")
    token = tree.means_token()
    means = tree.device_means()
    scalings = np.asarray(scalings, dtype=np.float64)
    if out == "torch" and order == "plan":
        counts = ctx.sample_counts(means, rows, scalings, np.asarray(alpha, dtype=np.float64), np.asarray(beta, dtype=np.float64),
                                   seed=seed, check_domain="deferred" if strict else False, means_token=token)
        if strict:
            ctx.domain_status()
        return counts
    # The cells are PRESENTED to the sampler grouped by their row of the mean tensor (every count is keyed by the cell's
    # position in the plan -- cell_index --, so the matrix is the same whatever the order): the kernel then finds a gene
    # tile's rows of the mean tensor in cache instead of fetching them once per cell, 2 to 5 % of its time
    # (profiles/r05_ablation.txt).  The device matrix is in the order of presentation; the copy to the host puts the rows
    # back in plan order chunk by chunk (a gather on the device, under the transfer of the previous chunk).
    perm = _device.plan_order(rows, means.shape[0])
    counts = ctx.sample_counts(means, rows[perm], scalings[perm], np.asarray(alpha, dtype=np.float64),
                               np.asarray(beta, dtype=np.float64), seed=seed, cell_index=perm.astype(np.int64),
                               check_domain="deferred" if strict else False, means_token=token)
    if out == "torch":
        if strict:
            ctx.domain_status()
        return _device.PresentedCounts(counts, perm)
    try:
        host = _host_return(counts, out, row_order=perm)
    except BaseException:
        if strict:
            _discard_verdict(ctx)       # (an OverflowError of "numpy16", a failed page-lock: the call's verdict must not outlive it)
        raise
    if strict:
        ctx.domain_status()
    return host


def _discard_verdict(ctx):
    """Read and clear the sticky verdict of deferred domain checks whose call is being abandoned."""
    try:
        ctx.domain_status()
    except (ValueError, RuntimeError):
        pass


def add_non_diff_genes(inform_expr_matrix, genes, gene_params, cell_scalings, *, seed=None):
    """Append ``genes`` non-differential genes (simulation.py:654-675): the same sampler
    with one constant mean row; returns float64 like the reference."""
    N, G = inform_expr_matrix.shape
    if seed is None:
        lo, hi = random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        seed = int(lo) | (int(hi) << 32)
    ctx = _device.get_context()
    base = np.asarray(gene_params["base_expr"], dtype=np.float64).reshape(1, genes)
    extra = ctx.sample_counts(base.astype(np.float32), np.zeros(N, dtype=np.int32),
                              np.asarray(cell_scalings, dtype=np.float64),
                              np.broadcast_to(np.asarray(gene_params["alpha"], dtype=np.float64), (genes,)),
                              np.broadcast_to(np.asarray(gene_params["beta"], dtype=np.float64), (genes,)),
                              seed=seed)
    fusion = np.zeros((N, G + genes))
    fusion[:, 0:G] = inform_expr_matrix
    fusion[:, G:] = extra.cpu().numpy()
    return fusion
