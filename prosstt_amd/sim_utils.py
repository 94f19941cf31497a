"""
Helpers of the simulation, mirroring ``prosstt.sim_utils``
(reference: /root/reference/prosstt/sim_utils.py).  Same names, arguments and
return values.  Everything O(branches) or O(cells) stays host Python/numpy and
issues the SAME numpy RandomState calls in the same order as the reference, so
seeded scripts see identical (pseudotime, branch, scaling, base-expression)
draws; everything O(cells x genes) or O(time x genes) is done by HIP kernels.
"""
import collections
import numbers
import operator
import sys
from collections import defaultdict, deque

import numpy as np
from numpy import random

from . import device as _device


def print_progress(iteration, total, prefix='', suffix='', decimals=1):
    """Terminal progress bar (sim_utils.py:22-49; never called by the library)."""
    width = 80
    percent = ("{0:." + str(decimals) + "f}").format(100 * (iteration / float(total)))
    filled = int(round(width * iteration / float(total)))
    sys.stdout.write('\r%s |%s| %s%s %s' % (prefix, '#' * filled + '-' * (width - filled), percent, '%', suffix))
    if iteration == total:
        sys.stdout.write('\n')
    sys.stdout.flush()


def random_partition(k, iterable):
    """Random partition into k groups, one ``randint`` per element (sim_utils.py:52-73)."""
    results = [[] for _ in range(k)]
    for value in iterable:
        results[random.randint(k)].append(value)
    return results


def test_correlation(W, k, cutoff):
    """sim_utils.py:76-94.  The reference loops over ``range(k - 1, 0)``, which is
    empty for every k >= 1, and for k == 0 compares row 0 with the still-zero last
    row (Pearson r = NaN): it never reports a correlation and consumes no random
    numbers.  Reproduced as the constant it is."""
    return False


def create_groups(no_programs, no_genes):
    """Two random module memberships per gene (sim_utils.py:97-126)."""
    first = random_partition(no_programs, random.permutation(no_genes))
    second = random_partition(no_programs, random.permutation(no_genes))
    return [a + b for a, b in zip(first, second)]


def bifurc_adjust(child, parent):
    """Shift ``child`` so that its first row equals the last row of ``parent`` (sim_utils.py:129-142)."""
    return child - (child[0] - parent[-1])


def pearson_between_programs(genes, prog1, prog2):
    """Per-gene Pearson r of two (T, genes) matrices over their common length
    (sim_utils.py:145-168), all genes at once; constant series give NaN as scipy does.
    Utility for users: ``simulate_lineage`` gets the sign counts from the fused
    ``lineage_attempt`` kernel instead."""
    common = min(prog1.shape[0], prog2.shape[0])
    x, y = np.asarray(prog1[:common], float), np.asarray(prog2[:common], float)
    xm, ym = x - x.mean(axis=0), y - y.mean(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = ((xm / np.sqrt((xm * xm).sum(axis=0))) * (ym / np.sqrt((ym * ym).sum(axis=0)))).sum(axis=0)
    r[np.all(x == x[0], axis=0) | np.all(y == y[0], axis=0)] = np.nan
    return np.clip(r, -1.0, 1.0)[:genes]


def flat_order(n):
    """Rows [index, i, j] of the upper triangle of an n x n matrix (sim_utils.py:171-187)."""
    res = np.zeros((int(n * (n - 1) / 2), 3), dtype=int)
    index = 0
    for i in range(n - 1):
        for j in range(i + 1, n):
            res[index] = (index, i, j)
            index += 1
    return res


def calc_relat_means(tree, programs, coefficients):
    """programs @ coefficients for every branch (sim_utils.py:190-213), on the device."""
    import torch
    ctx = _device.get_context()
    H = ctx.tensor(np.asarray(coefficients, dtype=np.float64), torch.float64)
    out = {}
    for branch in tree.branches:
        T = np.shape(programs[branch])[0]
        rel = torch.empty((T, H.shape[1]), dtype=torch.float64, device=ctx.torch_device)
        ctx.lineage_commit(programs[branch], H, rel, None)
        out[branch] = rel.cpu().numpy()
    return out


def diverging_parallel(branches, programs, genes, tol=0.5):
    """For every pair of parallel branches: do more than ``tol`` of the genes run in opposite
    directions, i.e. have a negative Pearson coefficient between the two branches' (T, genes)
    relative means (sim_utils.py:216-252)?  One answer per pair, in ``flat_order``'s order; a
    lone branch has nothing to diverge from."""
    present = [b for b in branches if b is not None]
    if len(present) == 1:
        return [True]
    verdict = []
    for _, i, j in flat_order(len(present)):
        signs = pearson_between_programs(genes, programs[present[i]], programs[present[j]])
        with np.errstate(invalid="ignore"):
            verdict.append(np.count_nonzero(signs < 0) / float(genes) > tol)
    return np.array(verdict, dtype=bool)


def commited_branches(tree, branches, rel_means):
    """Blend two sibling branches into each other over the first timezone they share
    (sim_utils.py:255-271).  The weight of the sibling falls linearly from just under 1/2 at the
    start of the zone to 0 at its end.  As in the reference, both branches are cut down to the
    zone's steps, the zone is located with the first branch's start on its left edge and the
    second branch's start on its right edge, and the second branch is blended with the ALREADY
    blended first one."""
    first, second = branches
    zones = tree.populate_timezone()
    owners = assign_branches(tree.branch_times(), zones)
    shared = min(i for i, who in enumerate(owners.values()) if who == branches)
    starts = tree.branch_times()
    lo = zones[shared][0] - starts[first][0]
    hi = zones[shared][1] - starts[second][0]
    steps = np.arange(lo, hi + 1)
    w_sibling = np.arange(0, 0.5, 1 / (2 * len(steps)))[::-1][:, None]
    w_own = 1 - w_sibling
    rel_means[first] = w_own * rel_means[first][steps] + w_sibling * rel_means[second][steps]
    rel_means[second] = w_own * rel_means[second][steps] + w_sibling * rel_means[first][steps]
    return rel_means


def assign_branches(branch_times, timezone):
    """Branches that contain each timezone (sim_utils.py:274-315)."""
    res = defaultdict(list)
    for i, zone in enumerate(timezone):
        for k in branch_times:
            if belongs_to(zone, branch_times[k]):
                res[i].append(k)
    return res


def belongs_to(timezone, branch):
    """sim_utils.py:318-339."""
    return (timezone[0] >= branch[0]) and (timezone[1] <= branch[1])


def pick_branches(tree, pseudotime, *, fix_density_index=False):
    """Random branch for every pseudotime value (sim_utils.py:342-403).

    One uniform per cell is consumed, exactly like the reference's per-cell
    ``random.choice(possibilities, p=probabilities)``, so the branch stream is
    identical; the work is grouped by (timezone, offset) instead of looping over
    cells.  The reference indexes ``tree.density[b]`` with the offset inside the
    *timezone* (:393-396); that is reproduced unless ``fix_density_index=True``.
    Unlike the reference (:361) labels are not truncated to the width of the first
    branch name."""
    pseudotime = np.asarray(pseudotime)
    timezone = tree.populate_timezone()
    assignments = assign_branches(tree.branch_times(), timezone)
    bt = tree.branch_times()
    labels = np.empty(len(pseudotime), dtype=object)
    uniforms = random.random_sample(len(pseudotime))
    zone_start = np.array([z[0] for z in timezone])
    zone_end = np.array([z[1] for z in timezone])
    zone_of = np.searchsorted(zone_start, pseudotime, side="right") - 1
    if np.any(zone_of < 0) or np.any(pseudotime > zone_end[np.clip(zone_of, 0, None)]):
        raise IndexError("pseudotime outside the tree")
    keys = zone_of.astype(np.int64) * (int(zone_end.max()) + 2) + (pseudotime - zone_start[zone_of])
    for key in np.unique(keys):
        cells = np.nonzero(keys == key)[0]
        z = int(zone_of[cells[0]])
        where = int(pseudotime[cells[0]] - timezone[z][0])
        possibilities = assignments[z]
        dens = np.zeros(len(possibilities))
        for i, b in enumerate(possibilities):
            at = where + (timezone[z][0] - bt[b][0]) if fix_density_index else where
            dens[i] = tree.density[b][at]
        cdf = (dens / dens.sum()).cumsum()
        cdf /= cdf[-1]
        picked = cdf.searchsorted(uniforms[cells], side="right")
        choices = np.empty(len(possibilities), dtype=object)
        choices[:] = possibilities
        labels[cells] = choices[picked]
    first = tree.branches[0]
    if isinstance(first, str):
        return labels.astype(str)
    return np.array(list(labels), dtype=np.asarray(tree.branches).dtype)


def pick_branch(tree, pseudotime, timezones, assignments):
    """One cell of ``pick_branches`` (sim_utils.py:367-403)."""
    zone = next(i for i, z in enumerate(timezones) if z[0] <= pseudotime <= z[1])
    possibilities = assignments[zone]
    where = pseudotime - timezones[zone][0]
    densities = np.array([tree.density[b][where] for b in possibilities])
    return random.choice(possibilities, p=densities / densities.sum())


def max_relat_exp(tree, relative_means):
    """(G, branches) maximum of exp(relative mean) per gene and branch (sim_utils.py:406-426).
    The per-branch maxima of the log values come from the device; exp is monotone."""
    import torch
    from . import simulation as sim
    ctx = _device.get_context()
    rel = sim._device_rel(tree, relative_means)
    offsets, _ = tree.row_offsets()
    maxes = np.zeros((tree.G, len(tree.branches)))
    for i, b in enumerate(tree.branches):
        gmax = torch.full((tree.G,), -np.inf, dtype=torch.float64, device=ctx.torch_device)
        ctx.gene_max(rel[offsets[b]:offsets[b] + int(tree.time[b])], gmax)
        maxes[:, i] = np.exp(gmax.cpu().numpy())
    return maxes


def simulate_base_gene_exp(tree, relative_means, abs_max=5000, gene_mean=0.8, gene_std=1,
                           *, max_attempts=None):
    """Base expression per gene: redraw exp(N(gene_mean, gene_std)) until
    ``value * max relative expression <= abs_max`` (sim_utils.py:429-470).

    The per-gene maximum over the whole tree is reduced on the device; the draws
    are the reference's sequential scalar normals (same stream, same order).
    ``max_attempts`` (new, default unlimited like the reference) bounds the redraws
    per gene."""
    from . import simulation as sim
    log_max = sim._device_gene_max(tree, relative_means).cpu().numpy()
    max_per_gene = np.exp(log_max)
    base_gene_exp = np.zeros(tree.G)
    for gene in range(tree.G):
        tmp = np.exp(random.standard_normal() * gene_std + gene_mean)
        tries = 0
        while tmp * max_per_gene[gene] > abs_max:
            tries += 1
            if max_attempts is not None and tries > max_attempts:
                raise RuntimeError("gene %d: no base expression below abs_max after %d draws" % (gene, tries))
            tmp = np.exp(random.standard_normal() * gene_std + gene_mean)
        base_gene_exp[gene] = tmp
    return base_gene_exp


def calc_scalings(cells, scale=True, scale_mean=0, scale_v=0.7):
    """Library-size factor per cell (sim_utils.py:473-498)."""
    if scale:
        return np.exp(random.standard_normal(cells) * scale_v + scale_mean)
    return np.ones(cells)


def process_timeseries_input(series_points, cells, point_std):
    """Arguments of sample_pseudotime_series as three arrays of one length (sim_utils.py:501-542).
    A total cell count is split evenly over the sample points (truncating); a scalar ``point_std``
    is DIVIDED by the number of sample points, as in the reference (:536-537)."""
    n_points = len(series_points)

    def per_point(value, dtype):
        if isinstance(value, collections.abc.Iterable):
            return np.array(value, dtype=dtype)
        if isinstance(value, numbers.Number):
            return np.full(n_points, value / n_points).astype(dtype)
        return value

    if not isinstance(series_points, np.ndarray):
        series_points = np.array(series_points, dtype=int)
    return series_points, per_point(cells, int), per_point(point_std, float)


def breadth_first_branches(tree):
    """Branches ordered by BFS level, ties in ``tree.branches`` order (sim_utils.py:545-567)."""
    levels = {branch: -1 for branch in tree.branches}
    levels[tree.root] = 0
    for parent, child in bfs_finder(np.array(tree.topology), tree.root):
        levels[child] = levels[parent] + 1
    ordered = sorted(levels.items(), key=operator.itemgetter(1))
    return np.array(ordered)[:, 0] if isinstance(tree.branches[0], str) else \
        np.array([b for b, _ in ordered])


def bfs_finder(graph, start):
    """Edges of ``graph`` ([parent, child] rows) in breadth-first order from ``start``
    (sim_utils.py:570-608)."""
    children = defaultdict(list)
    for parent, child in graph:
        children[parent].append(child)
    out, todo, done = [], deque([start]), set()
    while todo:
        node = todo.popleft()
        if node in done:
            continue
        done.add(node)
        for kid in children[node]:
            out.append([node, kid])
            todo.append(kid)
    return np.array(out).reshape(-1, 2) if out else np.empty((0, 2), dtype=graph.dtype)


def adjust_to_parent(relative_means, current, topology):
    """Shift branch ``current`` onto the end of its parent (sim_utils.py:611-640)."""
    topology = np.asarray(topology)
    if topology.size == 0:
        return relative_means[current]
    parent_loc = (topology[:, 1] == current)
    if not np.any(parent_loc):
        return relative_means[current]
    parent = topology[parent_loc][0][0]
    return bifurc_adjust(relative_means[current], relative_means[parent])


def find_parallel(tree, programs, branch):
    """Siblings of ``branch`` that already have programs, itself included (sim_utils.py:643-667)."""
    if len(tree.topology) > 0:
        for parallels in tree.get_parallel_branches().values():
            if branch in parallels:
                return np.intersect1d(parallels, list(programs.keys()))
    return [branch, None]


def learn_data_summary(cell_stats, gene_stats, relative_means):
    """Hyper-parameters that make a simulation resemble a real dataset's summaries
    (sim_utils.py:670-719): library-size distribution [mean, sd] of log(total / mean total);
    log alpha and log(beta - 1) from the weighted quadratic fit var ~ alpha*mean^2 + beta*mean + c
    over the genes with positive mean and variance; base expression = observed mean / average
    relative expression along the tree (floored at the smallest observed mean)."""
    totals = np.asarray(cell_stats.loc["total"], dtype=float)
    log_size = np.log(totals / totals.mean())
    observed_mean, observed_var = gene_stats.loc["means"], gene_stats.loc["var"]
    informative = (observed_var > 0) & (observed_mean > 0)
    quad, lin, _ = np.polyfit(observed_mean[informative], observed_var[informative], 2,
                              w=1 / observed_var[informative])
    along_tree = np.exp(np.array([relative_means[b] for b in relative_means.index]))   # (branches, T, G)
    avg_relative = along_tree.mean(axis=1).mean(axis=0)
    base = observed_mean[informative]
    avg_relative[avg_relative < base.min()] = base.min()
    return [log_size.mean(), log_size.std()], np.log(quad), np.log(lin - 1), np.array(base / avg_relative)
