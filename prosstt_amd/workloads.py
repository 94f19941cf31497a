"""
The benchmark / parity configurations of BASELINE.json, built through the product
pipeline (SURVEY.md section 8 d fixes topologies, sizes and seeds; the recipe is
examples/generate_simN.py:91-113 of the reference scaled up):

    C1  5-branch chain x 40 steps,        200 cells x   500 genes   (sample_whole_tree(t, 1))
    C2  single bifurcation (3 branches),   5k cells x    5k genes
    C3  8-branch tree,                    50k cells x   20k genes   <- the headline metric
    C4  32-branch tree,                  200k cells x   20k genes
    T32 the same 32-branch tree,          50k cells x   20k genes   <- north_star's target shape on ONE GPU
    C5  256-branch tree,                   1M cells x   30k genes

``np.random.seed(seed)`` precedes the topology; ``np.random.seed(seed + 1)`` precedes the
sampling plan, so the (pseudotime, branch, scaling) of every cell does not depend on how
much of the stream the lineage stage consumed.
"""
import time as _time

import numpy as np

from . import simulation as sim
from . import sim_utils as sut
from .tree import Tree

CONFIGS = {
    "C1": dict(kind="chain", branches=5, T=40, G=500, K=15, N=200, seed=92),
    "C2": dict(kind="bifurcation", T=50, G=5000, K=25, N=5000, seed=42),
    "C3": dict(kind="random", branch_points=3, T=50, G=20000, K=25, N=50000, seed=2024),
    "C4": dict(kind="random", branch_points=15, T=50, G=20000, K=25, N=200000, seed=2025),
    # north_star: ">= 40 % HBM roofline on 50k cells x 20k genes x 32-branch tree on 1 MI355X" -- C4's tree (same seed,
    # hence the same topology, lineage and mean tensor), C3's cell count
    "T32": dict(kind="random", branch_points=15, T=50, G=20000, K=25, N=50000, seed=2025),
    "C5": dict(kind="binary", depth=7, T=50, G=30000, K=25, N=1000000, seed=2026),
}


def topology_of(cfg):
    """Topology (list of [parent, child]) and branch labels of a configuration.
    Draws from the global numpy stream for kind='random' (Tree.gen_random_topology)."""
    kind = cfg["kind"]
    if kind == "chain":
        labels = [chr(ord("A") + i) for i in range(cfg["branches"])]
        return [[labels[i], labels[i + 1]] for i in range(len(labels) - 1)], labels
    if kind == "bifurcation":
        return [["A", "B"], ["A", "C"]], ["A", "B", "C"]
    if kind == "random":          # 2n+1 branches from the reference's generator, +1 chain branch
        top = [[int(a), int(b)] for a, b in Tree.gen_random_topology(cfg["branch_points"])]
        last = 2 * cfg["branch_points"]
        top.append([last, last + 1])
        return top, list(range(last + 2))
    if kind == "binary":          # balanced binary tree (2^(d+1)-1 branches) + 1 chain branch
        n = 2 ** (cfg["depth"] + 1) - 1
        top = [[(c - 1) // 2, c] for c in range(1, n)]
        top.append([n - 1, n])
        return top, list(range(n + 1))
    raise ValueError(kind)


class Workload:
    def __init__(self, name, cfg, tree, alpha, beta, info):
        self.name, self.cfg, self.tree, self.alpha, self.beta, self.info = name, cfg, tree, alpha, beta, info

    def plan(self, n_cells=None):
        """(pseudotime, branch labels, scalings, rows) of ``n_cells`` cells (default: the config's N).
        On a tree that is sharded over processes, ``rows`` is -1 for cells of branches held elsewhere."""
        n_cells = self.cfg["N"] if n_cells is None else n_cells
        np.random.seed(self.cfg["seed"] + 1)
        if self.name == "C1":
            pt, br = sim.cover_whole_tree(self.tree)
            pt, br = np.asarray(pt), np.asarray(br)
        else:
            pt, br = sim._density_plan(self.tree, n_cells)
        sc = sut.calc_scalings(len(pt))
        if self.tree._branch_owner is None:
            return pt, br, sc, sim.cell_rows(self.tree, pt, br)
        held = np.isin(br, np.asarray(self.tree.resident_branches()))
        rows = np.full(len(pt), -1, dtype=np.int32)
        rows[held] = sim.cell_rows(self.tree, pt[held], br[held])
        return pt, br, sc, rows


def build(name, *, a=0.05, rel_exp_cutoff=8, max_attempts=20000, G=None, verbose=False, group=None, sharded=None):
    """Tree with lineage, base expression and mean tensor on the device, plus (alpha, beta).
    Under an initialised process group of more than one rank the lineage is built sharded
    (``parallel.simulate_lineage_sharded``: gene slices for the attempts, every rank keeps the branches it
    owns) unless ``sharded=False``."""
    cfg = dict(CONFIGS[name])
    if G is not None:
        cfg["G"] = G
    np.random.seed(cfg["seed"])
    topology, labels = topology_of(cfg)
    tree = Tree(topology=topology, time={b: cfg["T"] for b in labels}, num_branches=len(labels),
                branch_points=len({p for p, _ in topology}), modules=cfg["K"], G=cfg["G"])
    stats = []
    t0 = _time.perf_counter()
    from . import parallel
    if sharded is None:
        sharded = parallel.world(group)[1] > 1
    if sharded:
        rel, _, _ = parallel.simulate_lineage_sharded(tree, rel_exp_cutoff, 0, 0, group=group, a=a,
                                                      max_attempts=max_attempts, stats=stats, keep_on_device=True)
    else:
        rel, _, _ = sim.simulate_lineage(tree, a=a, intra_branch_tol=0, inter_branch_tol=0,
                                         rel_exp_cutoff=rel_exp_cutoff, max_attempts=max_attempts, stats=stats,
                                         keep_on_device=True)       # (3 GB of binary64 at C5 stay where add_genes reads them)
    t1 = _time.perf_counter()
    base = sut.simulate_base_gene_exp(tree, rel)
    tree.add_genes(rel, base)
    tree.device_means()
    t2 = _time.perf_counter()
    alpha = np.exp(np.random.normal(np.log(0.2), np.log(1.5), cfg["G"]))      # generate_simN.py:94-95
    beta = np.exp(np.random.normal(np.log(1), np.log(1.5), cfg["G"])) + 1
    info = dict(branches=len(labels), rows=int(sum(tree.time.values)), attempts=len(stats),
                lineage_s=t1 - t0, base_and_means_s=t2 - t1, sharded=bool(sharded),
                resident_rows=int(tree.row_offsets()[1]))
    if verbose:
        print("[workload %s] %s" % (name, info))
    return Workload(name, cfg, tree, alpha, beta, info)
