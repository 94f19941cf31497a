#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by importing the REAL reference.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference, only the .npz files this script wrote.  The reference
publishes no tests or fixtures of its own, so these vectors are the
pin for every parity claim.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py
    PROSSTT_GOLDEN_OUT=/tmp/check python3 tests/golden/make_golden.py      # regenerate elsewhere, then compare

Everything stored is data: inputs (seeds, topologies, parameters) and the
reference's outputs.  No reference source text is stored.
"""
import json
import os
import sys
import types
import warnings

REF = os.environ.get("PROSSTT_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules["newick"] = types.ModuleType("newick")   # only Tree.from_newick needs it
warnings.filterwarnings("ignore")

import numpy as np                                     # noqa: E402
import scipy.stats                                     # noqa: E402
from prosstt import tree as rtree                      # noqa: E402
from prosstt import simulation as rsim                 # noqa: E402
from prosstt import sim_utils as rsut                  # noqa: E402
from prosstt import count_model as rcm                 # noqa: E402

OUT = os.environ.get("PROSSTT_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))   # (a scratch directory to re-check the committed files)

# The five topologies of SURVEY section 8(c) G3/G5 (+ an unequal-length one).
TREES = {
    "bifurcation": dict(topology=[["A", "B"], ["A", "C"]],
                        time={"A": 40, "B": 40, "C": 40}, branch_points=1),
    "chain6": dict(topology=[["A", "B"], ["B", "C"], ["C", "D"], ["D", "E"], ["E", "F"]],
                   time={b: 30 for b in "ABCDEF"}, branch_points=0),
    "chainbif7": dict(topology=[["A", "B"], ["B", "C"], ["B", "D"], ["D", "E"], ["C", "F"], ["F", "G"]],
                      time={b: 50 for b in "ABCDEFG"}, branch_points=1),
    "star5": dict(topology=[[0, 1], [0, 2], [0, 3], [0, 4], [0, 5]],
                  time={b: 20 for b in range(6)}, branch_points=1),
    "unequal": dict(topology=[["A", "B"], ["A", "C"], ["C", "D"], ["C", "E"]],
                    time={"A": 70, "B": 100, "C": 60, "D": 25, "E": 33}, branch_points=2),
}


def build_tree(spec, G, modules):
    return rtree.Tree(topology=spec["topology"], time=spec["time"],
                      num_branches=len(spec["time"]), branch_points=spec["branch_points"],
                      modules=modules, G=G)


def tree_json(spec, G, modules, extra=None):
    d = dict(topology=spec["topology"], time={str(k): int(v) for k, v in spec["time"].items()},
             int_labels=isinstance(next(iter(spec["time"])), int),
             branch_points=spec["branch_points"], G=G, modules=modules)
    d.update(extra or {})
    return json.dumps(d)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# --------------------------------------------------------------------- G1
def g1_get_pr_umi():
    m = np.concatenate([10.0 ** np.arange(-7, 5), [0.5, 1.5, 3.0, 12.0, 16.0, 250.0]])
    a = np.array([0.0, 1e-4, 0.05, 0.2, 1.0, 3.0])
    b = np.array([1.0 + 1e-8, 1.3, 2.0, 3.0, 8.0])
    A, B, M = [x.reshape(-1) for x in np.meshgrid(a, b, m, indexing="ij")]
    p, r = rcm.get_pr_umi(A, B, M)
    save("g1_get_pr_umi", a=A, b=B, m=M, p=p, r=r)


# --------------------------------------------------------------------- G2
def g2_walks():
    out = {}
    for seed in (0, 1, 92):
        for T in (2, 40, 50):
            np.random.seed(seed)
            out["diffusion_s%d_T%d" % (seed, T)] = rsim.diffusion(T)
    np.random.seed(7)
    out["sim_expr_branch_s7_T50_K5"] = rsim.sim_expr_branch(50, 5)
    save("g2_walks", **out)


# --------------------------------------------------------------------- G3
def g3_lineage():
    for name, spec in TREES.items():
        for mode, kwargs in (("gamma", dict(a=0.05)), ("beta", dict(a=2, b=2))):
            if mode == "beta" and name not in ("bifurcation", "star5"):
                continue
            G, K, seed = 64, 5, 1000 + len(name)
            np.random.seed(seed)
            t = build_tree(spec, G, K)
            log = []
            orig_pearson = rsut.pearson_between_programs
            orig_branch = rsim.sim_expr_branch

            def spy_pearson(genes, p1, p2):
                r = orig_pearson(genes, p1, p2)
                log[-1][1].append(int(np.sum(r < 0)))
                return r

            def spy_branch(*a, **k):
                log.append([None, []])
                return orig_branch(*a, **k)

            rsut.pearson_between_programs = spy_pearson
            rsim.sim_expr_branch = spy_branch
            try:
                rel, prog, H = rsim.simulate_lineage(t, intra_branch_tol=0, **kwargs)
            finally:
                rsut.pearson_between_programs = orig_pearson
                rsim.sim_expr_branch = orig_branch
            state_after = np.random.random_sample()      # pins total RNG consumption
            arrays = dict(tree=tree_json(spec, G, K, dict(seed=seed, kwargs=kwargs)),
                          H=H, attempts=len(log), state_after=state_after,
                          anticorr=np.array([json.dumps(c[1]) for c in log]),
                          bfs=np.array([str(b) for b in rsut.breadth_first_branches(t)]))
            for b in t.branches:
                arrays["rel_%s" % b] = rel[b]
                arrays["prog_%s" % b] = prog[b]
            save("g3_lineage_%s_%s" % (name, mode), **arrays)


# --------------------------------------------------------------------- G4
def g4_params():
    spec = TREES["bifurcation"]
    np.random.seed(11)
    t = build_tree(spec, 200, 6)
    rel, _, _ = rsim.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    np.random.seed(12)
    base = rsut.simulate_base_gene_exp(t, rel)
    np.random.seed(13)
    al, be = rcm.generate_negbin_params(t, mean_alpha=0.2, mean_beta=3)
    np.random.seed(14)
    sc = rsut.calc_scalings(300, True, 0.1, 0.7)
    sc_off = rsut.calc_scalings(5, False)
    arrays = dict(tree=tree_json(spec, 200, 6), base=base, alpha=al, beta=be,
                  scalings=sc, scalings_off=sc_off)
    for b in t.branches:
        arrays["rel_%s" % b] = rel[b]
    save("g4_params", **arrays)


# --------------------------------------------------------------------- G5
def g5_topology():
    out = {}
    for name, spec in TREES.items():
        t = build_tree(spec, 10, 3)
        bt = t.branch_times()
        zones = t.populate_timezone()
        rec = dict(
            branch_times={str(k): [int(x) for x in v] for k, v in bt.items()},
            timezone=[[int(x) for x in z] for z in zones],
            assign={str(i): [str(b) for b in v] for i, v in rsut.assign_branches(bt, zones).items()},
            bfs=[str(b) for b in rsut.breadth_first_branches(t)],
            parallel={str(k): [str(x) for x in v] for k, v in t.get_parallel_branches().items()},
            max_time=int(t.get_max_time()),
            density_sum=float(sum(np.sum(v) for v in t.density.values())),
            cover=[[int(x) for x in rsim.cover_whole_tree(t)[0]],
                   [str(x) for x in rsim.cover_whole_tree(t)[1]]],
        )
        out[name] = json.dumps(rec)
        out[name + "_tree"] = tree_json(spec, 10, 3)
    np.random.seed(5)
    out["random_topology_bp4_seed5"] = np.array(rtree.Tree.gen_random_topology(4))
    np.random.seed(2024)
    out["random_topology_bp3_seed2024"] = np.array(rtree.Tree.gen_random_topology(3))
    save("g5_topology", **out)


# --------------------------------------------------------------------- G6
def g6_sampling():
    for name, N, G, K in (("bifurcation", 60, 48, 5), ("unequal", 80, 40, 4), ("star5", 50, 32, 4)):
        spec = TREES[name]
        seed = 300 + N
        np.random.seed(seed)
        t = build_tree(spec, G, K)
        rel, prog, H = rsim.simulate_lineage(t, a=0.05, intra_branch_tol=0)
        base = rsut.simulate_base_gene_exp(t, rel)
        t.add_genes(rel, base)
        alpha = np.exp(np.random.normal(np.log(0.2), np.log(1.5), G))
        beta = np.exp(np.random.normal(np.log(1), np.log(1.5), G)) + 1
        np.random.seed(seed + 1)
        X, pt, br, sc = rsim.sample_density(t, N, alpha=alpha, beta=beta)
        # deterministic intermediates of draw_counts (simulation.py:633-645)
        bt = t.branch_times()
        mu = np.array([t.means[b][p - bt[b][0]] * s for p, b, s in zip(pt, br, sc)])
        pr = [rcm.get_pr_umi(alpha, beta, row) for row in mu]
        arrays = dict(tree=tree_json(spec, G, K, dict(seed=seed)), base=base, H=H,
                      alpha=alpha, beta=beta, X=X, pt=pt, br=np.array([str(b) for b in br]),
                      scalings=sc, mu=mu, p=np.array([x[0] for x in pr]),
                      r=np.array([x[1] for x in pr]))
        for b in t.branches:
            arrays["means_%s" % b] = t.means[b]
            arrays["rel_%s" % b] = rel[b]
        # the other samplers, same tree, fresh seeds
        np.random.seed(seed + 2)
        X2, pt2, br2, sc2 = rsim.sample_whole_tree(t, 2, alpha=alpha, beta=beta)
        arrays.update(wt_X=X2, wt_pt=np.array(pt2), wt_br=np.array([str(b) for b in br2]), wt_sc=sc2)
        np.random.seed(seed + 3)
        times = np.arange(0, t.get_max_time(), 3)
        X3, pt3, br3, sc3 = rsim._sample_data_at_times(t, times, alpha=alpha, beta=beta)
        arrays.update(at_X=X3, at_pt=pt3, at_br=np.array([str(b) for b in br3]), at_sc=sc3)
        np.random.seed(seed + 4)
        X4, pt4, br4, sc4 = rsim.sample_pseudotime_series(
            t, 30, [5, 30, 60], 6.0, alpha=0.3, beta=2)
        arrays.update(ps_X=X4, ps_pt=pt4, ps_br=np.array([str(b) for b in br4]), ps_sc=sc4)
        np.random.seed(seed + 5)
        gp = dict(alpha=alpha[:7], beta=beta[:7], base_expr=base[:7])
        arrays.update(nd_X=rsim.add_non_diff_genes(X, 7, gp, sc))
        save("g6_sampling_%s" % name, **arrays)


# --------------------------------------------------------------------- G7
def g7_nb_tables():
    """Analytic NB law for a grid of (m, a, b): pmf[0..kmax], mean, variance
    (scipy.stats.nbinom with the reference's parametrisation n=r, p=1-p)."""
    grid = [(0.05, 0.2, 2.0), (0.5, 0.2, 2.0), (1.5, 0.15, 2.2), (1.5, 1e-4, 1.5), (4.0, 0.3, 3.0),
            (9.0, 0.2, 2.0), (11.9, 1.0, 1.3), (12.1, 0.2, 2.0), (18.9, 0.2, 2.0), (18.9, 0.0, 1.0 + 1e-8),
            (19.1, 0.2, 2.0), (15.0, 0.9, 2.4), (15.0, 1.1, 2.4), (30.0, 0.2, 2.0), (30.0, 2.0, 2.0),
            (100.0, 0.05, 1.5), (100.0, 1.2, 4.0), (400.0, 0.2, 2.0), (3000.0, 0.25, 2.0),
            (3.0, 5.0, 2.0), (20.0, 0.0, 1.0 + 1e-8), (0.7, 0.0, 1.0 + 1e-8), (2.0, 0.0, 7.0),
            (8.0, 3.0, 1.0), (50.0, 1e-4, 1.2),
            # the inversion class reaches beyond m = 19 where the NB is overdispersed (-log P0 <= 19, theta <= 16 in PRNB-5)
            (60.0, 0.2, 2.0), (75.0, 0.1, 3.0), (80.0, 0.1, 3.0), (106.0, 0.0, 17.0), (40.0, 0.37, 2.0),
            (44.0, 0.37, 2.0), (25.0, 0.02, 1.3),
            # PRNB-6: the inversion class up to theta <= 24 -- its new corner (long walks), the middle of the new range, just outside by -log P0
            (135.0, 0.163, 2.0), (100.0, 0.2, 2.0), (140.0, 0.157, 2.0), (60.0, 0.38, 1.2),
            # round 6: tiny means, where the binary32 P(X = 0) is relatively weakest (and the Poisson limit)
            (1e-4, 0.2, 2.0), (1e-3, 0.2, 2.0), (1e-2, 0.2, 2.0), (1e-4, 0.0, 1.0 + 1e-8), (1e-3, 0.0, 1.0 + 1e-8),
            (1e-2, 0.0, 1.0 + 1e-8)]
    kmax = 4096
    k = np.arange(kmax)
    pmf = np.zeros((len(grid), kmax))
    par = np.zeros((len(grid), 7))
    for i, (m, a, b) in enumerate(grid):
        p, r = rcm.get_pr_umi(np.array([a]), np.array([b]), np.array([m]))
        d = scipy.stats.nbinom(n=r[0], p=1 - p[0])
        pmf[i] = d.pmf(k)
        par[i] = (m, a, b, r[0], p[0], d.mean(), d.var())
    save("g7_nb_tables", params=par, pmf=pmf.astype(np.float64))


# --------------------------------------------------------------------- G8
def g8_end_to_end():
    np.random.seed(92)
    t = rtree.Tree()
    X, pt, br, sc = rsim.sample_whole_tree_restricted(t)
    save("g8_minimal_example", X=X, pt=pt, br=np.array([str(b) for b in br]), scalings=sc,
         modules=t.modules, total=int(X.sum()))
    # config C1 of BASELINE.json: 5-branch chain x 40 steps, sample_whole_tree(t, 1) -> 200 x 500
    spec = dict(topology=[["A", "B"], ["B", "C"], ["C", "D"], ["D", "E"]],
                time={b: 40 for b in "ABCDE"}, branch_points=0)
    np.random.seed(92)
    t = build_tree(spec, 500, 15)
    rel, _, H = rsim.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    base = rsut.simulate_base_gene_exp(t, rel)
    t.add_genes(rel, base)
    alpha, beta = rcm.generate_negbin_params(t)
    np.random.seed(93)
    X, pt, br, sc = rsim.sample_whole_tree(t, 1, alpha=alpha, beta=beta)
    save("g8_config1_chain", tree=tree_json(spec, 500, 15), X=X.astype(np.int32), pt=np.array(pt),
         br=np.array([str(b) for b in br]), scalings=sc, alpha=alpha, beta=beta, base=base,
         H=H.astype(np.float64), rel_E=rel["E"], total=int(X.sum()))


# --------------------------------------------------------------------- G9
def g9_helpers():
    """commited_branches, process_timeseries_input, learn_data_summary (sim_utils.py:255-271, 501-542,
    670-719) on small inputs."""
    import pandas as pd
    out = {}
    rng = np.random.default_rng(9)
    # commited_branches: the two children of a bifurcation, blended over their common timezone
    spec = TREES["bifurcation"]
    t = build_tree(spec, 6, 4)
    rel = pd.Series({b: rng.normal(size=(spec["time"][b], 6)) for b in "ABC"})
    out.update({"cb_in_%s" % b: rel[b].copy() for b in "ABC"})
    res = rsut.commited_branches(t, ["B", "C"], rel.copy())
    out.update({"cb_out_%s" % b: np.asarray(res[b]) for b in "ABC"})
    # process_timeseries_input: scalar and list forms
    cases = [([5, 30, 60], 100, 6.0), ([10, 20], [30, 50], [2.0, 3.5]), (np.array([7, 8, 9, 10]), 33, [1.0, 2.0, 3.0, 4.0])]
    for i, (pts, cells, std) in enumerate(cases):
        a, b, c = rsut.process_timeseries_input(pts, cells, std)
        out.update({"ts%d_points" % i: a, "ts%d_cells" % i: b, "ts%d_std" % i: c})
    # learn_data_summary: summaries of a small synthetic count matrix
    G, N = 12, 40
    X = rng.negative_binomial(3, 0.3, size=(N, G)).astype(float)
    cell_stats = pd.DataFrame({"total": X.sum(axis=1), "zeros": (X == 0).sum(axis=1)}).T
    gene_stats = pd.DataFrame({"means": X.mean(axis=0), "var": X.var(axis=0), "zeros": (X == 0).sum(axis=0)}).T
    relm = pd.Series({b: rng.normal(scale=0.3, size=(15, G)) for b in "ABC"})
    scale, la, lb, prop = rsut.learn_data_summary(cell_stats, gene_stats, relm)
    out.update(ld_X=X, ld_scale=np.array(scale), ld_alpha=np.array(la), ld_beta=np.array(lb), ld_means=prop)
    out.update({"ld_rel_%s" % b: relm[b] for b in "ABC"})
    save("g9_helpers", **out)


# --------------------------------------------------------------------- G10
def g10_written_files():
    """Files written by the reference's tree_utils.save_* (tree_utils.py:59-173) for a small simulation,
    stored byte for byte together with the inputs that produced them."""
    import tempfile
    from prosstt import tree_utils as rtut
    rng = np.random.default_rng(10)
    spec = TREES["bifurcation"]
    t = build_tree(spec, 5, 3)
    N, G = 7, 5
    X = rng.integers(0, 40, size=(N, G)).astype(np.int64)
    labs = rng.integers(0, 80, N)
    brns = np.array(list("ABCABCA"))
    scal = np.exp(rng.normal(0, 0.7, N))
    alpha, beta, scale = np.exp(rng.normal(-1.6, 0.4, G)), np.exp(rng.normal(0, 0.4, G)) + 1, np.exp(rng.normal(0.8, 1, G))
    H = rng.gamma(0.05, size=(3, G))
    uMs = {b: rng.normal(size=(4, G)) for b in "ABC"}
    out = dict(X=X, labs=labs, brns=brns, scalings=scal, alpha=alpha, beta=beta, genescale=scale, H=H,
               tree=tree_json(spec, 5, 3), rseed=np.array(42))
    out.update({"uMs_%s" % b: uMs[b] for b in "ABC"})
    with tempfile.TemporaryDirectory() as d:
        rtut.save_cell_params("job", d, labs, brns, scal)
        rtut.save_gene_params("job", d, scale, alpha, beta)
        rtut.save_matrices("job", d, X, uMs, H)
        rtut.save_params("job", d, t, 42)
        for fn in sorted(os.listdir(d)):
            out["file_" + fn] = np.frombuffer(open(os.path.join(d, fn), "rb").read(), dtype=np.uint8)
    save("g10_written_files", **out)


# --------------------------------------------------------------------- G11
class _Node:
    """Stand-in for a node of the third-party ``newick`` package (absent here): exactly the attributes
    tree_utils.parse_newick reads -- name, length, descendants, ancestor, walk() in pre-order."""
    def __init__(self, name, length):
        self.name, self.length, self.descendants, self.ancestor = name, length, [], None

    def walk(self):
        yield self
        for child in self.descendants:
            yield from child.walk()


def _nodes_from_table(names, lengths, parents):
    nodes = [_Node(str(n), float(l)) for n, l in zip(names, lengths)]
    for node, parent in zip(nodes, parents):
        if parent >= 0:
            node.ancestor = nodes[parent]
            nodes[parent].descendants.append(node)
    return nodes


def g11_velocity_and_newick():
    """Tree.set_velocity -> density (tree.py:241-264, tree_utils.py:176-242; numpy 2 dropped the ``np.Inf``
    the reference spells, restored here as an alias) and tree_utils.parse_newick (tree_utils.py:10-56) on
    node tables (name, length, parent) that the test turns into the same stand-in nodes."""
    from prosstt import tree_utils as rtut
    np.Inf = np.inf
    out = {}
    rng = np.random.default_rng(11)
    for tname in ("bifurcation", "unequal"):
        spec = TREES[tname]
        for vname, lo in (("pos", 0.2), ("neg", -1.5)):        # the second set dips below zero: sanitize_velocity shifts it
            t = build_tree(spec, 4, 3)
            vel = {b: rng.uniform(lo, 3.0, size=spec["time"][b]) for b in spec["time"]}
            out.update({"vel_%s_%s_in_%s" % (tname, vname, b): vel[b].copy() for b in vel})
            t.set_velocity({b: vel[b].copy() for b in vel})
            out.update({"vel_%s_%s_density_%s" % (tname, vname, b): np.asarray(t.density[b]) for b in vel})
        out["vel_%s_tree" % tname] = tree_json(spec, 4, 3)
    # parse_newick: (name, length, parent index) tables in pre-order; 0 = "no length given" -> def_time,
    # fractional lengths are truncated by int()
    tables = {
        "bif": (["A", "B", "C"], [50, 30, 0], [-1, 0, 0], "(B:30,C)A:50;"),
        "deep": (["root", "x", "x1", "x2", "y", "y1", "y1a", "y1b", "z"], [0, 12.7, 5, 0, 40, 8.2, 3, 3.9, 60],
                 [-1, 0, 1, 1, 0, 4, 5, 5, 0], "((x1:5,x2)x:12.7,((y1a:3,y1b:3.9)y1:8.2)y:40,z:60)root;"),
        "single": (["only"], [17], [-1], "only:17;"),
    }
    for name, (names, lengths, parents, text) in tables.items():
        nodes = _nodes_from_table(names, lengths, parents)
        top, time, branches, bpoints, root = rtut.parse_newick([nodes[0]], 40)
        out.update({"nw_%s_names" % name: np.array(names), "nw_%s_lengths" % name: np.array(lengths, float),
                    "nw_%s_parents" % name: np.array(parents), "nw_%s_text" % name: np.array(text),
                    "nw_%s_topology" % name: np.array(top).reshape(-1, 2) if top else np.zeros((0, 2), "<U1"),
                    "nw_%s_time_keys" % name: np.array(list(time.keys())),
                    "nw_%s_time_vals" % name: np.array(list(time.values())),
                    "nw_%s_counts" % name: np.array([branches, bpoints]),
                    "nw_%s_root" % name: np.array("" if root is None else root)})
    save("g11_velocity_newick", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # regenerate selected fixtures only, e.g. `make_golden.py g7_nb_tables`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_get_pr_umi()
    g2_walks()
    g3_lineage()
    g4_params()
    g5_topology()
    g6_sampling()
    g7_nb_tables()
    g9_helpers()
    g10_written_files()
    g11_velocity_and_newick()
    g8_end_to_end()
