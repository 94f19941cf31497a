"""
Pins oracle/ref_numpy.py (the numpy restatement of the reference CPU path)
against the golden vectors that tests/golden/make_golden.py wrote from the REAL
reference.  Every comparison here is bit-exact: same numpy primitives, same
order, same global RandomState.
"""
import json

import numpy as np
import pytest

from conftest import load_golden, tree_spec, label_of
from oracle import ref_numpy as ref

TREE_NAMES = ["bifurcation", "chain6", "chainbif7", "star5", "unequal"]


def make_tree(spec):
    return ref.RefTree(spec["topology"], spec["time"], num_branches=len(spec["time"]),
                       branch_points=spec["branch_points"], modules=spec["modules"], G=spec["G"])


def test_g1_get_pr_umi():
    g = load_golden("g1_get_pr_umi")
    p, r = ref.get_pr_umi(g["a"], g["b"], g["m"])
    np.testing.assert_array_equal(p, g["p"])
    np.testing.assert_array_equal(r, g["r"])


def test_g2_walks():
    g = load_golden("g2_walks")
    for seed in (0, 1, 92):
        for T in (2, 40, 50):
            np.random.seed(seed)
            np.testing.assert_array_equal(ref.diffusion(T), g["diffusion_s%d_T%d" % (seed, T)])
    np.random.seed(7)
    np.testing.assert_array_equal(ref.sim_expr_branch(50, 5), g["sim_expr_branch_s7_T50_K5"])


@pytest.mark.parametrize("name,mode", [(n, "gamma") for n in TREE_NAMES] +
                         [("bifurcation", "beta"), ("star5", "beta")])
def test_g3_lineage(name, mode):
    g = load_golden("g3_lineage_%s_%s" % (name, mode))
    spec = tree_spec(g)
    np.random.seed(spec["seed"])
    t = make_tree(spec)
    log = []
    rel, prog, H = ref.simulate_lineage(t, intra_branch_tol=0, log=log, **spec["kwargs"])
    assert np.random.random_sample() == float(g["state_after"])   # same RNG consumption
    np.testing.assert_array_equal(H, g["H"])
    assert len(log) == int(g["attempts"])
    assert [json.dumps(rec[2]) for rec in log] == list(g["anticorr"])
    assert [str(b) for b in ref.breadth_first_branches(t)] == list(g["bfs"])
    for b in t.branches:
        np.testing.assert_array_equal(prog[b], g["prog_%s" % b])
        np.testing.assert_array_equal(rel[b], g["rel_%s" % b])


def test_g4_params():
    g = load_golden("g4_params")
    spec = tree_spec(g)
    t = make_tree(spec)
    rel = {b: g["rel_%s" % b] for b in t.branches}
    np.random.seed(12)
    np.testing.assert_array_equal(ref.simulate_base_gene_exp(t, rel), g["base"])
    np.random.seed(13)
    al, be = ref.generate_negbin_params(t, mean_alpha=0.2, mean_beta=3)
    np.testing.assert_array_equal(al, g["alpha"])
    np.testing.assert_array_equal(be, g["beta"])
    np.random.seed(14)
    np.testing.assert_array_equal(ref.calc_scalings(300, True, 0.1, 0.7), g["scalings"])
    np.testing.assert_array_equal(ref.calc_scalings(5, False), g["scalings_off"])


@pytest.mark.parametrize("name", TREE_NAMES)
def test_g5_topology(name):
    g = load_golden("g5_topology")
    rec = json.loads(str(g[name]))
    spec = tree_spec(str(g[name + "_tree"]))
    t = make_tree(spec)
    bt = t.branch_times()
    assert {str(k): list(map(int, v)) for k, v in bt.items()} == rec["branch_times"]
    zones = t.populate_timezone()
    assert [[int(x) for x in z] for z in zones] == rec["timezone"]
    assign = ref.assign_branches(bt, zones)
    assert {str(i): [str(b) for b in v] for i, v in assign.items()} == rec["assign"]
    assert [str(b) for b in ref.breadth_first_branches(t)] == rec["bfs"]
    assert {str(k): [str(x) for x in v] for k, v in t.get_parallel_branches().items()} == rec["parallel"]
    assert t.get_max_time() == rec["max_time"]
    assert abs(sum(np.sum(v) for v in t.density.values()) - rec["density_sum"]) < 1e-15
    pt, br = ref.cover_whole_tree(t)
    assert [int(x) for x in pt] == rec["cover"][0]
    assert [str(x) for x in br] == rec["cover"][1]


def test_g5_random_topology():
    g = load_golden("g5_topology")
    np.random.seed(5)
    np.testing.assert_array_equal(np.array(ref.gen_random_topology(4)), g["random_topology_bp4_seed5"])
    np.random.seed(2024)
    np.testing.assert_array_equal(np.array(ref.gen_random_topology(3)), g["random_topology_bp3_seed2024"])


@pytest.mark.parametrize("name", ["bifurcation", "unequal", "star5"])
def test_g6_sampling(name):
    g = load_golden("g6_sampling_%s" % name)
    spec = tree_spec(g)
    seed = spec["seed"]
    # full pipeline from the seed, as the generator ran it
    np.random.seed(seed)
    t = make_tree(spec)
    rel, prog, H = ref.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    base = ref.simulate_base_gene_exp(t, rel)
    t.add_genes(rel, base)
    np.testing.assert_array_equal(H, g["H"])
    np.testing.assert_array_equal(base, g["base"])
    for b in t.branches:
        np.testing.assert_array_equal(t.means[b], g["means_%s" % b])
    alpha = np.exp(np.random.normal(np.log(0.2), np.log(1.5), spec["G"]))
    beta = np.exp(np.random.normal(np.log(1), np.log(1.5), spec["G"])) + 1
    np.testing.assert_array_equal(alpha, g["alpha"])
    N = len(g["pt"])

    np.random.seed(seed + 1)
    X, pt, br, sc = ref.sample_density(t, N, alpha=alpha, beta=beta)
    np.testing.assert_array_equal(pt, g["pt"])
    assert [str(b) for b in br] == list(g["br"])
    np.testing.assert_array_equal(sc, g["scalings"])
    assert X.dtype == np.int64
    np.testing.assert_array_equal(X, g["X"])
    mu, p, r = ref.nb_parameters(t, pt, br, sc, alpha, beta)
    np.testing.assert_array_equal(mu, g["mu"])
    np.testing.assert_array_equal(p, g["p"])
    np.testing.assert_array_equal(r, g["r"])

    np.random.seed(seed + 2)
    X2, pt2, br2, sc2 = ref.sample_whole_tree(t, 2, alpha=alpha, beta=beta)
    np.testing.assert_array_equal(X2, g["wt_X"])
    np.testing.assert_array_equal(np.array(pt2), g["wt_pt"])
    assert [str(b) for b in br2] == list(g["wt_br"])
    np.testing.assert_array_equal(sc2, g["wt_sc"])

    np.random.seed(seed + 3)
    times = np.arange(0, t.get_max_time(), 3)
    X3, pt3, br3, sc3 = ref.sample_data_at_times(t, times, alpha=alpha, beta=beta)
    assert [str(b) for b in br3] == list(g["at_br"])
    np.testing.assert_array_equal(sc3, g["at_sc"])
    np.testing.assert_array_equal(X3, g["at_X"])

    np.random.seed(seed + 4)
    X4, pt4, br4, sc4 = ref.sample_pseudotime_series(t, 30, [5, 30, 60], 6.0, alpha=0.3, beta=2)
    np.testing.assert_array_equal(pt4, g["ps_pt"])
    assert [str(b) for b in br4] == list(g["ps_br"])
    np.testing.assert_array_equal(X4, g["ps_X"])

    np.random.seed(seed + 5)
    gp = dict(alpha=alpha[:7], beta=beta[:7], base_expr=base[:7])
    nd = ref.add_non_diff_genes(X, 7, gp, sc)
    assert nd.dtype == np.float64
    np.testing.assert_array_equal(nd, g["nd_X"])


def test_g8_minimal_example():
    g = load_golden("g8_minimal_example")
    np.random.seed(92)
    t = ref.RefTree([["A", "B"], ["A", "C"]], {"A": 40, "B": 40, "C": 40}, 3, 1, None, 500)
    assert t.modules == int(g["modules"])
    X, pt, br, sc = ref.sample_whole_tree_restricted(t)
    assert X.shape == (80, 500) and int(X.sum()) == int(g["total"]) == 162404
    np.testing.assert_array_equal(X, g["X"])
    np.testing.assert_array_equal(pt, g["pt"])
    assert [str(b) for b in br] == list(g["br"])
    np.testing.assert_array_equal(sc, g["scalings"])


def test_g8_config1_chain():
    """BASELINE.json configs[0]: 5-branch chain x 40 steps, 200 cells x 500 genes."""
    g = load_golden("g8_config1_chain")
    spec = tree_spec(g)
    np.random.seed(92)
    t = make_tree(spec)
    rel, _, H = ref.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    base = ref.simulate_base_gene_exp(t, rel)
    t.add_genes(rel, base)
    alpha, beta = ref.generate_negbin_params(t)
    np.random.seed(93)
    X, pt, br, sc = ref.sample_whole_tree(t, 1, alpha=alpha, beta=beta)
    assert X.shape == (200, 500)
    np.testing.assert_array_equal(X, g["X"])
    np.testing.assert_array_equal(rel["E"], g["rel_E"])
    assert int(X.sum()) == int(g["total"])


def test_zero_mean_raises_like_scipy():
    """simulation.py:647-648: scipy's argument check rejects an exact-zero mean."""
    t = ref.RefTree([["A", "B"]], {"A": 3, "B": 3}, 2, 0, 2, 4)
    t.add_genes({"A": np.ones((3, 4)), "B": np.zeros((3, 4))})
    with pytest.raises(ValueError):
        ref.draw_counts(t, np.array([4]), ["B"], np.ones(1), [0.2] * 4, [2.0] * 4)
