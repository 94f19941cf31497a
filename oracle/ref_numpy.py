"""
ORACLE -- test infrastructure only.  NOT part of the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  ``prosstt_amd`` never does.

A from-scratch numpy restatement of the PROSSTT CPU path (reference:
soedinglab/prosstt v1.2.0).  Every function names the reference lines it
follows.  The reference draws every variate from numpy's *global legacy*
``RandomState`` (scipy ``rvs`` with ``random_state=None`` forwards to it), so a
restatement that issues the same primitive numpy calls in the same order is
**bit-identical** to the reference at equal ``np.random.seed``.  That is what
this file does, and ``tests/test_oracle_golden.py`` pins it against fixtures
written by ``tests/golden/make_golden.py`` from the real reference.

Third-party arithmetic on the path (numpy ``RandomState`` / scipy.stats, both
unpinned by the reference: setup.py:12) is reached here through numpy itself;
``oracle/numpy_legacy.c`` restates numpy's legacy negative-binomial chain in C
for a numpy-free check of the same stream.
"""
from collections import deque

import numpy as np

# --------------------------------------------------------------------------
# Tree container (reference: prosstt/tree.py:19-80) -- only what the path reads
# --------------------------------------------------------------------------


class RefTree:
    """Minimal lineage-tree record.  tree.py:51-80."""

    def __init__(self, topology, time, num_branches=None, branch_points=None,
                 modules=None, G=500, density=None, root=None):
        self.topology = [list(p) for p in topology]
        self.time = dict(time)
        self.branches = list(time.keys())
        self.num_branches = len(self.branches) if num_branches is None else num_branches
        if branch_points is None:
            branch_points = len({p[0] for p in self.topology})
        self.branch_points = branch_points
        self.G = G
        self.means = None
        if modules is None:
            # tree.py:67-68 consumes one randint from the global stream
            modules = 5 * branch_points + np.random.randint(1, 20)
        self.modules = modules
        self.root = self.branches[0] if root is None else root
        self.density = self.default_density() if density is None else density

    # tree.py:138-151
    def default_density(self):
        total = 0
        for b in self.branches:
            total += self.time[b]
        return {b: np.array([1. / total] * int(self.time[b])) for b in self.branches}

    # tree.py:376-399
    def branch_times(self):
        bt = {self.root: [0, self.time[self.root] - 1]}
        for parent, child in self.topology:
            end = bt[parent][1]
            bt[child] = [end + 1, end + self.time[child]]
        return bt

    # tree.py:425-434
    def get_parallel_branches(self):
        top = np.array(self.topology)
        return {b: top[top[:, 0] == b, 1] for b in np.unique(top[:, 0])}

    # tree.py:287-330
    def children(self):
        kids = {}
        for parent, child in self.topology:
            kids.setdefault(parent, []).append(child)
        return kids

    def paths(self, start):
        kids = self.children()
        if not kids.get(start):
            return [[start]]
        out = []
        for node in kids[start]:
            for tail in self.paths(node):
                out.append([start] + tail)
        return out

    # tree.py:267-285
    def get_max_time(self):
        return int(max(np.sum([self.time[b] for b in p]) for p in self.paths(self.root)))

    # tree.py:332-374, 402-423
    def populate_timezone(self):
        res = []
        stacks = []
        for path in self.paths(self.root):
            prev = 0
            stack = []
            for b in path:
                stack.append([prev, prev + int(self.time[b])])
                prev += int(self.time[b])
            stacks.append(stack)
        while stacks:
            starts = np.array([s[0][0] for s in stacks])
            ends = np.array([s[0][1] for s in stacks])
            if np.all(ends == np.max(ends)):
                res.append([np.max(starts), np.max(ends) - 1])
                for s in stacks:
                    s.pop(0)
            else:
                lo = np.min(ends)
                res.append([np.max(starts), lo - 1])
                for s in stacks:
                    if s[0][1] != lo:
                        s.insert(1, [lo, s[0][1]])
                    s.pop(0)
            stacks = [s for s in stacks if s]
        return res

    # tree.py:154-213
    def add_genes(self, relative_means, base_gene_expr=None):
        if base_gene_expr is None:
            avg = relative_means
        else:
            avg = {b: np.exp(relative_means[b]) * base_gene_expr for b in self.branches}
        if len(avg) != self.num_branches:
            raise ValueError("number of arrays must equal number of branches")
        for b in avg:
            if avg[b].shape != (self.time[b], self.G):
                raise ValueError("branch %s: expected %s, got %s"
                                 % (b, (self.time[b], self.G), avg[b].shape))
        self.means = avg


# tree.py:82-113
def gen_random_topology(branch_points, branch_names=None):
    total = 2 * branch_points + 1
    seeds = [0]
    avail = list(reversed(range(1, total)))
    if branch_names is None:
        branch_names = np.arange(total)
    res = []
    while avail:
        root = np.random.choice(seeds)
        a = avail.pop()
        b = avail.pop()
        res.append([branch_names[root], branch_names[a]])
        res.append([branch_names[root], branch_names[b]])
        seeds.append(a)
        seeds.append(b)
        seeds.remove(root)
    return res


# --------------------------------------------------------------------------
# Lineage stage (simulation.py:21-286, sim_utils.py helpers)
# --------------------------------------------------------------------------

def diffusion(steps):
    """simulation.py:89-124.  RNG order: U, N, U, (steps-1) x N."""
    walk = np.zeros(steps)
    vel = np.zeros(steps)
    walk[0] = np.log(np.random.random_sample() * 1.5 + 0)
    vel[0] = np.random.standard_normal() * 0.2 + 0
    s_eps = 2 / steps
    eta = np.random.random_sample() * 1 + 0
    eps = np.random.standard_normal(steps - 1) * s_eps + 0 if steps > 1 else ()
    for t in range(steps - 1):
        walk[t + 1] = walk[t] + vel[t]
        vel[t + 1] = eta * vel[t] + eps[t]
    return walk


def sim_expr_branch(branch_length, expr_progr):
    """simulation.py:21-86.  The correlation test (sim_utils.py:76-94) is a
    no-op for K>=2 (``range(k-1, 0)`` is empty), consumes no RNG, and therefore
    ``cutoff``/``max_loops`` have no effect; K==1 never terminates in the
    reference and is refused here."""
    if expr_progr < 2:
        raise ValueError("reference never terminates for a single program")
    programs = np.zeros((expr_progr, branch_length))
    for k in range(expr_progr):
        programs[k] = diffusion(branch_length)
    return np.transpose(programs)


def random_partition(k, iterable):
    """sim_utils.py:52-73."""
    out = [[] for _ in range(k)]
    for value in iterable:
        out[np.random.randint(k)].append(value)
    return out


def create_groups(no_programs, no_genes):
    """sim_utils.py:97-126."""
    g1 = random_partition(no_programs, np.random.permutation(no_genes))
    g2 = random_partition(no_programs, np.random.permutation(no_genes))
    return [x + y for x, y in zip(g1, g2)]


def simulate_coefficients(tree, **kwargs):
    """simulation.py:127-212.  'a' only -> Gamma(a); 'a' and 'b' -> Beta(2,2)
    on two random groups per gene (the passed values are ignored: :157-159);
    neither -> Gamma(0.04) (the reference also warns)."""
    K, G = tree.modules, tree.G
    if "a" not in kwargs:
        return np.reshape(np.random.standard_gamma(0.04, K * G) * 1 + 0, (K, G))
    if "b" in kwargs:
        groups = create_groups(K, G)
        H = np.zeros((K, G))
        for k in range(K):
            for gene in groups[k]:
                H[k][gene] += np.random.beta(2, 2) * 1 + 0
        return H
    return np.reshape(np.random.standard_gamma(kwargs["a"], K * G) * 1 + 0, (K, G))


def breadth_first_branches(tree):
    """sim_utils.py:545-608: BFS level per branch, then a *stable* sort of
    ``tree.branches`` by level (unreached branches keep level -1)."""
    kids = tree.children()
    level = {b: -1 for b in tree.branches}
    level[tree.root] = 0
    todo = deque([tree.root])
    seen = set()
    while todo:
        b = todo.popleft()
        if b in seen:
            continue
        seen.add(b)
        for c in kids.get(b, []):
            level[c] = level[b] + 1
            todo.append(c)
    return [b for b, _ in sorted(level.items(), key=lambda kv: kv[1])]


def parent_of(tree, branch):
    """sim_utils.py:632-635: first topology row whose child is ``branch``."""
    for parent, child in tree.topology:
        if child == branch:
            return parent
    return None


def adjust_to_parent(programs, branch, tree):
    """sim_utils.py:611-640 + 129-142: child - (child[0] - parent[-1])."""
    parent = parent_of(tree, branch)
    if parent is None:
        return programs[branch]
    dif = programs[branch][0] - programs[parent][-1]
    return programs[branch] - dif


def find_parallel(tree, programs, branch):
    """sim_utils.py:643-667."""
    for sibs in tree.get_parallel_branches().values():
        if branch in sibs:
            return list(np.intersect1d(sibs, list(programs.keys())))
    return [branch, None]


def pearson_columns(x, y):
    """Per-column Pearson r of two (T,G) arrays: scipy.stats.pearsonr's formula
    (centre, normalise, dot) applied to all genes at once (sim_utils.py:145-168).
    Constant columns give NaN, as scipy does."""
    xm = x - x.mean(axis=0)
    ym = y - y.mean(axis=0)
    nx = np.sqrt((xm * xm).sum(axis=0))
    ny = np.sqrt((ym * ym).sum(axis=0))
    with np.errstate(invalid="ignore", divide="ignore"):
        r = ((xm / nx) * (ym / ny)).sum(axis=0)
    const = np.all(x == x[0], axis=0) | np.all(y == y[0], axis=0)
    r[const] = np.nan
    return np.clip(r, -1.0, 1.0)


def anticorrelated_counts(parallels, rel_means, genes):
    """Number of genes with r<0 for every sibling pair i<j (sim_utils.py:216-252,
    pair order = flat_order 171-187)."""
    branches = [b for b in parallels if b is not None]
    out = []
    for i in range(len(branches) - 1):
        for j in range(i + 1, len(branches)):
            a, b = rel_means[branches[i]], rel_means[branches[j]]
            common = min(a.shape[0], b.shape[0])
            with np.errstate(invalid="ignore"):
                r = pearson_columns(a[:common], b[:common])
                out.append(int(np.sum(r < 0)))
    return out


def simulate_lineage(tree, rel_exp_cutoff=8, intra_branch_tol=0.5,
                     inter_branch_tol=0, log=None, **kwargs):
    """simulation.py:215-286.  Returns dicts keyed by branch label (the
    reference wraps them in pd.Series) and the (K,G) coefficient matrix.
    ``log`` (optional list) receives one record per attempt:
    (branch, max_rel, [anticorr count per sibling pair])."""
    if len(tree.time) != tree.num_branches:
        raise ValueError("the parameters are not enough for %i branches" % tree.num_branches)
    H = simulate_coefficients(tree, **kwargs)
    programs, rel_means = {}, {}
    for branch in breadth_first_branches(tree):
        while True:
            programs[branch] = sim_expr_branch(tree.time[branch], tree.modules)
            programs[branch] = adjust_to_parent(programs, branch, tree)
            rel_means[branch] = np.dot(programs[branch], H)
            mx = np.max(rel_means[branch])
            counts = anticorrelated_counts(find_parallel(tree, programs, branch),
                                           rel_means, tree.G)
            if log is not None:
                log.append((branch, float(mx), counts))
            diverges = all(c / (tree.G * 1.0) > inter_branch_tol for c in counts)
            if not (mx > rel_exp_cutoff) and diverges:
                break
    return rel_means, programs, H


def simulate_base_gene_exp(tree, relative_means, abs_max=5000, gene_mean=0.8, gene_std=1):
    """sim_utils.py:406-470: per gene, redraw exp(N(mean,std)) until
    tmp * max_b max_t exp(rel) <= abs_max."""
    maxes = np.zeros((tree.G, len(tree.branches)))
    for i, b in enumerate(tree.branches):
        maxes[:, i] = np.max(np.exp(relative_means[b]), axis=0)
    max_per_gene = np.max(maxes, axis=1)
    base = np.zeros(tree.G)
    for g in range(tree.G):
        tmp = np.exp(np.random.standard_normal() * gene_std + gene_mean)
        while tmp * max_per_gene[g] > abs_max:
            tmp = np.exp(np.random.standard_normal() * gene_std + gene_mean)
        base[g] = tmp
    return base


# --------------------------------------------------------------------------
# Count model (count_model.py:14-48, 131-161)
# --------------------------------------------------------------------------

def generate_negbin_params(tree, mean_alpha=0.2, mean_beta=2, a_scale=1.5, b_scale=1.5):
    """count_model.py:14-48 (log(scale) is used as a *std*, :43,45)."""
    alphas = np.exp(np.random.standard_normal(tree.G) * np.log(a_scale) + np.log(mean_alpha))
    betas = np.exp(np.random.standard_normal(tree.G) * np.log(b_scale) + np.log(mean_beta)) + 1
    return alphas, betas


def get_pr_umi(a, b, m):
    """count_model.py:131-161."""
    a, b, m = np.asarray(a, dtype=float), np.asarray(b, dtype=float), np.asarray(m, dtype=float)
    with np.errstate(divide="ignore", invalid="ignore"):
        s2 = a * m ** 2 + b * m
        p = (s2 - m) / s2
        r = (m ** 2) / (s2 - m)
    p = np.array(p, dtype=float)
    r = np.array(r, dtype=float)
    p[s2 <= 0] = 0
    r[s2 <= 0] = 0
    return p, r


# --------------------------------------------------------------------------
# Sampling stage (simulation.py:416-651, sim_utils.py:473-498)
# --------------------------------------------------------------------------

def calc_scalings(cells, scale=True, scale_mean=0, scale_v=0.7):
    """sim_utils.py:473-498."""
    if scale:
        return np.exp(np.random.standard_normal(cells) * scale_v + scale_mean)
    return np.ones(cells)


def cell_rows(tree, pseudotime, branches):
    """simulation.py:634-635: time index of each cell inside its branch."""
    bt = tree.branch_times()
    offsets = np.array([bt[b][0] for b in branches])
    return np.asarray(pseudotime) - offsets


def cell_means(tree, pseudotime, branches, scalings):
    """simulation.py:633-640: (N,G) float64 mean of every cell."""
    times = cell_rows(tree, pseudotime, branches)
    mu = np.zeros((len(branches), tree.G))
    for n, (t, b) in enumerate(zip(times, branches)):
        mu[n] = tree.means[b][t] * scalings[n]
    return mu


def nb_parameters(tree, pseudotime, branches, scalings, alpha, beta):
    """simulation.py:633-645: the deterministic (mu, p, r) of every count."""
    mu = cell_means(tree, pseudotime, branches, scalings)
    p = np.zeros_like(mu)
    r = np.zeros_like(mu)
    for n in range(mu.shape[0]):
        p[n], r[n] = get_pr_umi(alpha, beta, mu[n])
    return mu, p, r


def draw_counts(tree, pseudotime, branches, scalings, alpha, beta):
    """simulation.py:602-651.  ``scipy.stats.nbinom(n, p).rvs()`` forwards to
    ``RandomState.negative_binomial(n, p, size)`` of the global state and casts
    to int64; scipy's argument check rejects n<=0 or p outside (0,1]."""
    _, p, r = nb_parameters(tree, pseudotime, branches, scalings, alpha, beta)
    n_arg = r.reshape(-1)
    p_arg = 1 - p.reshape(-1)
    ok = (n_arg > 0) & (p_arg > 0) & (p_arg <= 1)
    if not np.all(ok):
        raise ValueError("Domain error in arguments.")
    x = np.random.negative_binomial(n_arg, p_arg, n_arg.shape)
    return x.astype(np.int64).reshape((len(branches), tree.G))


def draw_counts_as_reference(tree, pseudotime, branches, scalings, alpha, beta):
    """simulation.py:602-651 with the reference's own loop structure and library calls -- the CPU
    baseline of bench.py.  ``draw_counts`` above returns the same matrix faster (it looks the branch
    offsets up once); this one pays what the reference pays: ``tree.branch_times()`` rebuilt for
    every cell from a pandas Series of branch lengths (:634, tree.py:376-399), one row product per
    cell (:638-639), one ``get_pr_umi`` per cell on float64 rows written into the flat p/r vectors
    (:641-644), and ``scipy.stats.nbinom(n, p).rvs()`` on N*G elements (:647-648).
    tools/cpu_port_vs_reference.py times it against the imported reference."""
    import collections
    import pandas as pd
    import scipy.stats
    lengths = pd.Series({b: tree.time[b] for b in tree.branches})

    def branch_times():
        spans = collections.defaultdict(list)
        spans[tree.root] = [0, lengths[tree.root] - 1]
        for parent, child in tree.topology:
            end = spans[parent][1]
            spans[child] = [end + 1, end + lengths[child]]
        return spans

    n_cells, G = len(branches), tree.G
    mu = np.zeros((n_cells, G))
    starts = [branch_times()[b][0] for b in branches]
    steps = pseudotime - starts
    p_flat = np.zeros(n_cells * G)
    r_flat = np.zeros(n_cells * G)
    for n, t, b in zip(np.arange(n_cells), steps, branches):
        mu[n] = tree.means[b][t] * scalings[n]
    for n in range(n_cells):
        a, bb, m = alpha, beta, mu[n]
        s2 = (a * m ** 2 + bb * m)
        p = (s2 - m) / s2
        r = (m ** 2) / (s2 - m)
        p[s2 <= 0] = 0
        r[s2 <= 0] = 0
        p_flat[n * G:(n + 1) * G] = p
        r_flat[n * G:(n + 1) * G] = r
    return scipy.stats.nbinom(n=r_flat, p=(1 - p_flat)).rvs().reshape((n_cells, G))


def sample_data_at_times(tree, sample_pt, branches=None, alpha=0.3, beta=2,
                         scale=True, scale_mean=0., scale_v=0.7):
    """simulation.py:551-599."""
    if np.shape(alpha) == ():
        alpha = [alpha] * tree.G
    if np.shape(beta) == ():
        beta = [beta] * tree.G
    if branches is None:
        branches = pick_branches(tree, sample_pt)
    scalings = calc_scalings(len(sample_pt), scale, scale_mean, scale_v)
    X = draw_counts(tree, sample_pt, branches, scalings, alpha, beta)
    return X, sample_pt, branches, scalings


def density_plan(tree, no_cells):
    """simulation.py:452-467: the (pseudotime, branch) pair of every cell."""
    bt = tree.branch_times()
    pts = np.concatenate([np.arange(bt[b][0], bt[b][1] + 1) for b in tree.branches])
    labels = np.concatenate([[b] * tree.time[b] for b in tree.branches])
    prob = np.concatenate([tree.density[b] for b in tree.branches])
    sample = np.random.choice(np.arange(len(prob)), size=no_cells, p=prob)
    return pts[sample], labels[sample]


def sample_density(tree, no_cells, alpha=0.3, beta=2, scale=True, scale_v=0.7, scale_mean=0.):
    """simulation.py:416-471."""
    pt, br = density_plan(tree, no_cells)
    return sample_data_at_times(tree, pt, branches=br, alpha=alpha, beta=beta,
                                scale=scale, scale_mean=scale_mean, scale_v=scale_v)


# ---- index generators either side of the path (SURVEY section 8 f) ---------

def assign_branches(branch_times, timezone):
    """sim_utils.py:274-339."""
    res = {}
    for i, zone in enumerate(timezone):
        res[i] = [k for k, bt in branch_times.items() if zone[0] >= bt[0] and zone[1] <= bt[1]]
    return res


def pick_branches(tree, pseudotime):
    """sim_utils.py:342-403, including the reference's quirk of indexing
    ``density[b]`` with the offset inside the *timezone* (:393-396)."""
    zones = tree.populate_timezone()
    assignments = assign_branches(tree.branch_times(), zones)
    out = np.array([tree.branches[0]] * len(pseudotime))
    for n, t in enumerate(pseudotime):
        z = next(i for i, zone in enumerate(zones) if zone[0] <= t <= zone[1])
        poss = assignments[z]
        where = t - zones[z][0]
        dens = np.array([tree.density[b][where] for b in poss])
        out[n] = np.random.choice(poss, p=dens / dens.sum())
    return out


def cover_whole_tree(tree):
    """simulation.py:520-548."""
    zones = tree.populate_timezone()
    assignments = assign_branches(tree.branch_times(), zones)
    pt, br = [], []
    for i, (start, end) in enumerate(zones):
        for b in assignments[i]:
            pt.extend(np.arange(start, end + 1))
            br.extend([b] * (end + 1 - start))
    return pt, br


def sample_whole_tree(tree, n_factor, alpha=0.3, beta=2, scale=True, scale_mean=0., scale_v=0.7):
    """simulation.py:474-517."""
    pt, br = cover_whole_tree(tree)
    return sample_data_at_times(tree, np.repeat(pt, n_factor), branches=np.repeat(br, n_factor),
                                alpha=alpha, beta=beta, scale=scale,
                                scale_mean=scale_mean, scale_v=scale_v)


def default_gene_expression(tree):
    """tree.py:436-446."""
    rel, _, _ = simulate_lineage(tree, a=0.05)
    base = simulate_base_gene_exp(tree, rel)
    tree.add_genes({b: np.exp(rel[b]) * base for b in tree.branches})


def sample_whole_tree_restricted(tree, alpha=0.2, beta=3):
    """simulation.py:289-316 (returns 4 values, not the documented 3)."""
    sample_time = np.arange(0, tree.get_max_time())
    default_gene_expression(tree)
    alphas, betas = generate_negbin_params(tree, mean_alpha=alpha, mean_beta=beta)
    return sample_data_at_times(tree, sample_time, alpha=alphas, beta=betas)


def draw_times(timepoint, no_cells, max_time, var=4):
    """simulation.py:382-413."""
    pt = (np.random.standard_normal(no_cells) * var + timepoint).astype(int)
    pt[pt < 0] = 0
    pt[pt >= max_time] = max_time - 1
    return pt


def process_timeseries_input(series_points, cells, point_std):
    """sim_utils.py:501-542 (a scalar ``point_std`` is divided by the number
    of sample points, :536-537 -- reproduced)."""
    n = len(series_points)
    if np.ndim(cells) > 0:
        cells = np.array(cells, dtype=int)
    else:
        cells = np.array([cells / n] * n, dtype=int)
    if np.ndim(point_std) > 0:
        point_std = np.array(point_std, dtype=float)
    else:
        point_std = np.array([point_std / n] * n, dtype=float)
    return np.asarray(series_points, dtype=int), cells, point_std


def sample_pseudotime_series(tree, cells, series_points, point_std, alpha=0.3, beta=2,
                             scale=True, scale_mean=0, scale_v=0.7):
    """simulation.py:319-379."""
    series_points, cells, point_std = process_timeseries_input(series_points, cells, point_std)
    max_time = tree.get_max_time()
    pts = []
    for t, n, var in zip(series_points, cells, point_std):
        pts.extend(draw_times(t, n, max_time, var))
    return sample_data_at_times(tree, np.array(pts), alpha=alpha, beta=beta, scale=scale,
                                scale_mean=scale_mean, scale_v=scale_v)


def add_non_diff_genes(inform_expr_matrix, genes, gene_params, cell_scalings):
    """simulation.py:654-675 (returns float64)."""
    N, G = inform_expr_matrix.shape
    p_tot = np.zeros(N * genes)
    r_tot = np.zeros(N * genes)
    for c in range(N):
        p, r = get_pr_umi(gene_params["alpha"], gene_params["beta"],
                          cell_scalings[c] * gene_params["base_expr"])
        p_tot[c * genes:(c + 1) * genes] = p
        r_tot[c * genes:(c + 1) * genes] = r
    x = np.random.negative_binomial(r_tot, 1 - p_tot, r_tot.shape).astype(np.int64)
    out = np.zeros((N, G + genes))
    out[:, :G] = inform_expr_matrix
    out[:, G:] = x.reshape((N, genes))
    return out
