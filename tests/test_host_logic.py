"""
CPU tests (-m "not gpu") of the product's HOST side: the Tree container, the index
and RNG-order logic of prosstt_amd.simulation / sim_utils / count_model, checked
bit-exact against the golden vectors of the real reference, and the C ABI surface
(the library loads and exports every symbol include/prosstt_amd.h declares).
No compute call is made: without a GPU the numeric entry points must raise.
"""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, tree_spec

from prosstt_amd import tree as ptree
from prosstt_amd import simulation as sim
from prosstt_amd import sim_utils as sut
from prosstt_amd import count_model as cm
from prosstt_amd import tree_utils as tu
from prosstt_amd import _native

TREE_NAMES = ["bifurcation", "chain6", "chainbif7", "star5", "unequal"]


def make_tree(spec, G=None, modules=None):
    return ptree.Tree(topology=spec["topology"], time=spec["time"], num_branches=len(spec["time"]),
                      branch_points=spec["branch_points"], modules=modules or spec["modules"],
                      G=G or spec["G"])


# ---- C ABI surface -------------------------------------------------------------

def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "prosstt_amd.h")).read()
    declared = set(re.findall(r"\b(prosstt_amd_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_native.SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.load().prosstt_amd_version() == 600       # PRNB-7 (the version moves with the sampler's definition)


def test_no_cpu_fallback():
    """Without a GPU every numeric entry point fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    t = ptree.Tree(G=20, modules=3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sim.simulate_lineage(t, a=0.05)
    t.means = {b: np.ones((40, 20)) for b in t.branches}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sim.draw_counts(t, np.array([0]), ["A"], np.ones(1), [0.2] * 20, [2.0] * 20)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cm.sample_counts(np.ones((2, 3)), 0.2, 2.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        t.add_genes({b: np.zeros((40, 20)) for b in t.branches}, np.ones(20))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "prosstt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("ORACLE", ""), os.path.join(dirpath, f)


# ---- Tree container vs reference (fixture g5) --------------------------------------

@pytest.mark.parametrize("name", TREE_NAMES)
def test_tree_helpers(name):
    g = load_golden("g5_topology")
    rec = json.loads(str(g[name]))
    t = make_tree(tree_spec(str(g[name + "_tree"])))
    bt = t.branch_times()
    assert {str(k): [int(x) for x in v] for k, v in bt.items()} == rec["branch_times"]
    zones = t.populate_timezone()
    assert [[int(x) for x in z] for z in zones] == rec["timezone"]
    assign = sut.assign_branches(bt, zones)
    assert {str(i): [str(b) for b in v] for i, v in assign.items()} == rec["assign"]
    assert [str(b) for b in sut.breadth_first_branches(t)] == rec["bfs"]
    assert {str(k): [str(x) for x in v] for k, v in t.get_parallel_branches().items()} == rec["parallel"]
    assert t.get_max_time() == rec["max_time"]
    assert abs(sum(np.sum(v) for v in t.density.values()) - rec["density_sum"]) < 1e-15
    pt, br = sim.cover_whole_tree(t)
    assert [int(x) for x in pt] == rec["cover"][0]
    assert [str(x) for x in br] == rec["cover"][1]
    offsets, rows = t.row_offsets()
    assert rows == sum(t.time.values) and list(offsets) == t.branches


def test_random_topology_and_default_modules():
    g = load_golden("g5_topology")
    np.random.seed(5)
    np.testing.assert_array_equal(np.array(ptree.Tree.gen_random_topology(4)), g["random_topology_bp4_seed5"])
    np.random.seed(2024)
    np.testing.assert_array_equal(np.array(ptree.Tree.gen_random_topology(3)), g["random_topology_bp3_seed2024"])
    np.random.seed(92)
    assert ptree.Tree().modules == int(load_golden("g8_minimal_example")["modules"])


def test_tree_validation():
    with pytest.raises(ValueError):     # child listed before it exists as a branch end
        ptree.Tree(topology=[["B", "C"], ["A", "B"]], time={"A": 5, "B": 5, "C": 5}, num_branches=3, modules=2)
    t = ptree.Tree(G=7, modules=3)
    with pytest.raises(ValueError):
        t.add_genes({"A": np.ones((40, 7)), "B": np.ones((40, 7))})
    with pytest.raises(ValueError):
        t.add_genes({"A": np.ones((40, 7)), "B": np.ones((40, 7)), "C": np.ones((39, 7))})
    with pytest.raises(ValueError):
        t.set_density({"A": np.ones(40)})
    t.add_genes({b: np.full((40, 7), 2.0) for b in t.branches})
    assert t.means["B"].shape == (40, 7)
    with pytest.raises(ValueError):
        sim.simulate_lineage(ptree.Tree(num_branches=4, G=5, modules=2), a=0.05)


def test_velocity_to_density():
    t = ptree.Tree(G=5, modules=2)
    t.set_velocity({b: np.linspace(-1, 2, 40) for b in t.branches})
    total = sum(np.sum(v) for v in t.density.values())
    assert abs(total - 1) < 1e-12 and all(np.all(v > 0) for v in t.density.values())


# ---- RNG-order host draws vs reference -----------------------------------------------

def test_walks_bit_exact():
    g = load_golden("g2_walks")
    for seed in (0, 1, 92):
        for T in (2, 40, 50):
            np.random.seed(seed)
            np.testing.assert_array_equal(sim.diffusion(T), g["diffusion_s%d_T%d" % (seed, T)])
    np.random.seed(7)
    np.testing.assert_array_equal(sim.sim_expr_branch(50, 5), g["sim_expr_branch_s7_T50_K5"])
    with pytest.raises(ValueError):
        sim.sim_expr_branch(10, 1)


@pytest.mark.parametrize("name,mode", [("bifurcation", "gamma"), ("star5", "beta")])
def test_coefficients_bit_exact(name, mode):
    g = load_golden("g3_lineage_%s_%s" % (name, mode))
    spec = tree_spec(g)
    np.random.seed(spec["seed"])
    t = make_tree(spec)
    H = sim.simulate_coefficients(t, **spec["kwargs"])
    np.testing.assert_array_equal(H, g["H"])
    with pytest.warns(UserWarning):
        sim.simulate_coefficients(t)


def test_first_root_attempt_programs_bit_exact():
    """The first attempt of the root consumes the stream right after the coefficients."""
    g = load_golden("g3_lineage_chain6_gamma")
    spec = tree_spec(g)
    if int(g["attempts"]) != len(spec["time"]):
        pytest.skip("fixture has retries; covered on the GPU")
    np.random.seed(spec["seed"])
    t = make_tree(spec)
    sim.simulate_coefficients(t, **spec["kwargs"])
    programs = {}
    topology = np.array(t.topology)
    for b in sut.breadth_first_branches(t):
        programs[b] = sim.sim_expr_branch(t.time[b], t.modules)
        programs[b] = sut.adjust_to_parent(programs, b, topology)
        np.testing.assert_array_equal(programs[b], g["prog_%s" % b])


def test_params_and_scalings_bit_exact():
    g = load_golden("g4_params")
    t = make_tree(tree_spec(g))
    np.random.seed(13)
    al, be = cm.generate_negbin_params(t, mean_alpha=0.2, mean_beta=3)
    np.testing.assert_array_equal(al, g["alpha"])
    np.testing.assert_array_equal(be, g["beta"])
    np.random.seed(14)
    np.testing.assert_array_equal(sut.calc_scalings(300, True, 0.1, 0.7), g["scalings"])
    np.testing.assert_array_equal(sut.calc_scalings(5, False), g["scalings_off"])
    g1 = load_golden("g1_get_pr_umi")
    p, r = cm.get_pr_umi(g1["a"], g1["b"], g1["m"])
    np.testing.assert_array_equal(p, g1["p"])
    np.testing.assert_array_equal(r, g1["r"])


@pytest.mark.parametrize("name", ["bifurcation", "unequal", "star5"])
def test_sampling_plans_bit_exact(name):
    """(pseudotime, branch, scaling) of every sampler: integer assignments bit-exact."""
    g = load_golden("g6_sampling_%s" % name)
    spec = tree_spec(g)
    t = make_tree(spec)
    seed, N = spec["seed"], len(g["pt"])

    np.random.seed(seed + 1)
    pt, br = sim._density_plan(t, N)
    sc = sut.calc_scalings(N)
    np.testing.assert_array_equal(pt, g["pt"])
    assert [str(b) for b in br] == list(g["br"])
    np.testing.assert_array_equal(sc, g["scalings"])
    rows = sim.cell_rows(t, pt, br)
    offsets, _ = t.row_offsets()
    bt = t.branch_times()
    want = [offsets[type(t.branches[0])(b) if not isinstance(t.branches[0], str) else b] + p -
            bt[type(t.branches[0])(b) if not isinstance(t.branches[0], str) else b][0] for p, b in zip(pt, br)]
    np.testing.assert_array_equal(rows, np.array(want, dtype=np.int32))
    with pytest.raises(IndexError):
        sim.cell_rows(t, pt + 1000, br)

    np.random.seed(seed + 3)
    times = np.arange(0, t.get_max_time(), 3)
    picked = sut.pick_branches(t, times)
    assert [str(b) for b in picked] == list(g["at_br"])
    np.testing.assert_array_equal(sut.calc_scalings(len(times)), g["at_sc"])

    np.random.seed(seed + 4)
    series, cells, std = sut.process_timeseries_input([5, 30, 60], 30, 6.0)
    pts = np.concatenate([sim.draw_times(a, n, t.get_max_time(), v) for a, n, v in zip(series, cells, std)])
    np.testing.assert_array_equal(pts, g["ps_pt"])
    assert [str(b) for b in sut.pick_branches(t, pts)] == list(g["ps_br"])

    pt2, br2 = sim.cover_whole_tree(t)
    np.testing.assert_array_equal(np.repeat(pt2, 2), g["wt_pt"])
    assert [str(b) for b in np.repeat(br2, 2)] == list(g["wt_br"])


def test_pick_branch_single_matches_vector():
    g = load_golden("g6_sampling_unequal")
    t = make_tree(tree_spec(g))
    zones = t.populate_timezone()
    assign = sut.assign_branches(t.branch_times(), zones)
    times = np.arange(0, t.get_max_time(), 7)
    np.random.seed(3)
    one = [sut.pick_branch(t, p, zones, assign) for p in times]
    np.random.seed(3)
    assert [str(b) for b in sut.pick_branches(t, times)] == [str(b) for b in one]


def test_text_outputs(tmp_path):
    X = np.arange(6).reshape(2, 3)
    tu.save_matrices("job", str(tmp_path), X, {"A": np.ones((2, 3))}, np.eye(2))
    tu.save_cell_params("job", str(tmp_path), [0, 1], ["A", "A"], [1.0, 2.0])
    tu.save_gene_params("job", str(tmp_path), [1, 2, 3], [0.1, 0.2, 0.3], [2, 2, 2])
    tu.save_params("job", str(tmp_path), ptree.Tree(G=3, modules=2), 7)
    lines = open(tmp_path / "job_simulation.txt").read().splitlines()
    assert lines[0].split("\t") == ["", "gene_0", "gene_1", "gene_2"] and lines[2].split("\t") == ["cell_1", "3", "4", "5"]
    assert np.loadtxt(tmp_path / "job_umsA.txt").shape == (2, 3)
    assert "Genes: 3" in open(tmp_path / "job_params.txt").read()


def test_install_as_prosstt():
    import sys
    import prosstt_amd
    saved = {k: v for k, v in sys.modules.items() if k == "prosstt" or k.startswith("prosstt.")}
    try:
        prosstt_amd.install_as_prosstt()
        from prosstt import simulation as s2, tree as t2
        assert s2 is sim and t2 is ptree
    finally:
        for k in [k for k in sys.modules if k == "prosstt" or k.startswith("prosstt.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_from_newick():
    """Tree.from_newick (tree.py:115-126, tree_utils.py:10-56): pre-order topology, lengths as
    branch times (0 / missing -> def_time), root = the node without ancestor."""
    t = ptree.Tree.from_newick("((C:30,D:20)B:50,E)A:40;", modules=4, genes=12)
    assert t.topology == [["A", "B"], ["A", "E"], ["B", "C"], ["B", "D"]]
    assert dict(t.time) == {"A": 40, "B": 50, "C": 30, "D": 20, "E": 40}
    assert t.root == "A" and t.num_branches == 5 and t.branch_points == 2 and t.G == 12
    assert dict(t.branch_times()) == {"A": [0, 39], "B": [40, 89], "E": [40, 79], "C": [90, 119], "D": [90, 109]}
    assert [str(b) for b in sut.breadth_first_branches(t)] == ["A", "B", "E", "C", "D"]
    with pytest.raises(ValueError):
        ptree.Tree.from_newick("((C,D)B,E;")


def test_helpers_against_reference_fixtures():
    """commited_branches, process_timeseries_input, learn_data_summary against outputs of the
    reference (fixture g9, tests/golden/make_golden.py)."""
    import pandas as pd
    from prosstt_amd import sim_utils as sut
    from prosstt_amd.tree import Tree
    g = load_golden("g9_helpers")
    t = Tree(topology=[["A", "B"], ["A", "C"]], time={"A": 40, "B": 40, "C": 40}, num_branches=3,
             branch_points=1, modules=4, G=6)
    rel = pd.Series({b: g["cb_in_%s" % b].copy() for b in "ABC"})
    out = sut.commited_branches(t, ["B", "C"], rel)
    for b in "ABC":
        np.testing.assert_array_equal(np.asarray(out[b]), g["cb_out_%s" % b])
    cases = [([5, 30, 60], 100, 6.0), ([10, 20], [30, 50], [2.0, 3.5]), (np.array([7, 8, 9, 10]), 33, [1.0, 2.0, 3.0, 4.0])]
    for i, (pts, cells, std) in enumerate(cases):
        a, b, c = sut.process_timeseries_input(pts, cells, std)
        for got, key in ((a, "points"), (b, "cells"), (c, "std")):
            want = g["ts%d_%s" % (i, key)]
            assert got.dtype == want.dtype
            np.testing.assert_array_equal(got, want)
    X = g["ld_X"]
    cell_stats = pd.DataFrame({"total": X.sum(axis=1), "zeros": (X == 0).sum(axis=1)}).T
    gene_stats = pd.DataFrame({"means": X.mean(axis=0), "var": X.var(axis=0), "zeros": (X == 0).sum(axis=0)}).T
    relm = pd.Series({b: g["ld_rel_%s" % b] for b in "ABC"})
    scale, la, lb, prop = sut.learn_data_summary(cell_stats, gene_stats, relm)
    np.testing.assert_allclose(scale, g["ld_scale"], rtol=1e-12)
    np.testing.assert_allclose(la, g["ld_alpha"], rtol=1e-10)
    np.testing.assert_allclose(lb, g["ld_beta"], rtol=1e-10)
    np.testing.assert_allclose(prop, g["ld_means"], rtol=1e-12)


def test_writers_reproduce_the_reference_files(tmp_path):
    """tree_utils.save_* write, byte for byte, the files the reference's writers wrote for the same
    inputs (fixture g10: tree_utils.py:59-173); the binary writer keeps int32 and round-trips."""
    from prosstt_amd import tree_utils as tut
    g = load_golden("g10_written_files")
    spec = tree_spec(g)
    t = ptree.Tree(topology=spec["topology"], time=spec["time"], num_branches=len(spec["time"]),
                   branch_points=spec["branch_points"], modules=spec["modules"], G=spec["G"])
    uMs = {b: g["uMs_%s" % b] for b in "ABC"}
    d = str(tmp_path)
    tut.save_cell_params("job", d, g["labs"], g["brns"], g["scalings"])
    tut.save_gene_params("job", d, g["genescale"], g["alpha"], g["beta"])
    tut.save_matrices("job", d, g["X"], uMs, g["H"])
    tut.save_params("job", d, t, int(g["rseed"]))
    names = sorted(k[5:] for k in g.files if k.startswith("file_"))
    assert sorted(os.listdir(d)) == names
    for fn in names:
        assert open(os.path.join(d, fn), "rb").read() == bytes(g["file_" + fn]), fn
    path = tut.save_matrices_npz("job", d, g["X"], uMs, g["H"])
    back = np.load(path)
    assert back["X"].dtype == np.int32
    np.testing.assert_array_equal(back["X"], g["X"])
    np.testing.assert_array_equal(back["umsB"], uMs["B"])
    with pytest.raises(ValueError):
        tut.save_matrices_npz("job", d, np.array([[2 ** 40]]))
    # a sparse matrix (draw_counts(..., out="csr")) is stored in scipy's own layout, the other arrays beside it
    import scipy.sparse as sparse
    path = tut.save_matrices_npz("sparse", d, sparse.csr_matrix(g["X"]), uMs, g["H"])
    again = sparse.load_npz(path)
    assert again.format == "csr" and again.dtype == np.int32
    np.testing.assert_array_equal(again.toarray(), g["X"])
    np.testing.assert_array_equal(np.load(path)["H"], g["H"])


def test_host_arrays_stay_writable_and_edits_reach_the_device(monkeypatch):
    """The caller's mean arrays are never frozen; ``tree.means[b] = array`` and an in-place edit both
    invalidate the device copy (a fingerprint is compared), an untouched dict does not re-upload."""
    from prosstt_amd import device

    class Fake:
        torch_device = "cpu"
        uploads = 0

        def tensor(self, host, dtype):
            import torch
            Fake.uploads += 1
            return torch.as_tensor(np.array(host)).to(dtype)

    monkeypatch.setattr(device, "get_context", lambda *a, **k: Fake())
    t = ptree.Tree(topology=[["A", "B"]], time={"A": 4, "B": 3}, num_branches=2, branch_points=0, modules=2, G=5)
    mine = {"A": np.full((4, 5), 2.0), "B": np.full((3, 5), 3.0)}
    t.means = mine
    d0 = t.device_means()
    assert Fake.uploads == 1 and t.device_means() is d0 and Fake.uploads == 1
    assert all(a.flags.writeable for a in mine.values())            # caller-owned arrays are left alone
    mine["A"][1, 2] = 7.0                                            # in-place edit of the caller's array
    d1 = t.device_means()
    assert Fake.uploads == 2 and float(d1[1, 2]) == 7.0
    t.means["B"] = np.full((3, 5), 9.0)                              # the reference's idiom: item assignment
    d2 = t.device_means()
    assert Fake.uploads == 3 and float(d2[4, 0]) == 9.0 and float(d2[1, 2]) == 7.0
    assert t.device_means() is d2 and Fake.uploads == 3
    # fingerprints: identity and content both count, non-contiguous arrays are fine
    a = np.arange(12.0).reshape(3, 4)
    f = device.host_fingerprint([a, a[:, ::2]])
    assert f == device.host_fingerprint([a, a[:, ::2]])[:1] + f[1:] and f[0][3] != f[1][3]
    a[2, 3] += 1
    assert device.host_fingerprint([a])[0] != f[0]
    assert device.host_fingerprint([a.copy()])[0][0] != device.host_fingerprint([a])[0][0]


def test_velocity_and_newick_against_reference_fixtures():
    """Tree.set_velocity -> density and tree_utils.parse_newick against outputs of the reference
    (fixture g11: tree.py:241-264, tree_utils.py:10-56, 176-242).  parse_newick is fed the stand-in
    nodes the fixture describes AND the product's own Newick reader on the same text."""
    from prosstt_amd import _newick
    g = load_golden("g11_velocity_newick")
    for tname in ("bifurcation", "unequal"):
        spec = tree_spec(str(g["vel_%s_tree" % tname]))
        for vname in ("pos", "neg"):
            t = ptree.Tree(topology=spec["topology"], time=spec["time"], num_branches=len(spec["time"]),
                           branch_points=spec["branch_points"], modules=spec["modules"], G=spec["G"])
            vel = {b: g["vel_%s_%s_in_%s" % (tname, vname, b)].copy() for b in t.branches}
            if vname == "neg":
                assert min(v.min() for v in vel.values()) < 0
            t.set_velocity(vel)
            for b in t.branches:
                # bit for bit: np.random.choice builds its cdf from these numbers (the cells of a plan hang on the last bit)
                np.testing.assert_array_equal(t.density[b], g["vel_%s_%s_density_%s" % (tname, vname, b)])
    for name in ("bif", "deep", "single"):
        nodes = [_newick.Node(str(n), float(l)) for n, l in zip(g["nw_%s_names" % name], g["nw_%s_lengths" % name])]
        for node, parent in zip(nodes, g["nw_%s_parents" % name]):
            if parent >= 0:
                nodes[int(parent)].add(node)
        for parsed in ([nodes[0]], _newick.loads(str(g["nw_%s_text" % name]))):
            top, time, branches, bpoints, root = tu.parse_newick(parsed, 40)
            assert [list(map(str, row)) for row in g["nw_%s_topology" % name]] == top
            assert list(time.keys()) == [str(k) for k in g["nw_%s_time_keys" % name]]
            assert list(time.values()) == [int(v) for v in g["nw_%s_time_vals" % name]]
            assert [branches, bpoints] == g["nw_%s_counts" % name].tolist()
            assert (root or "") == str(g["nw_%s_root" % name])


def test_plan_order_is_a_stable_grouping_by_row():
    """prosstt_amd_plan_order (host helper of the C ABI, no device): the order simulation.draw_counts presents its cells in."""
    from prosstt_amd import device, _native
    rng = np.random.default_rng(0)
    for n, rows in ((0, 5), (1, 1), (1000, 7), (5000, 1600)):
        roc = rng.integers(0, rows, n).astype(np.int32)
        assert np.array_equal(device.plan_order(roc, rows), np.argsort(roc, kind="stable"))
    with pytest.raises(_native.NativeError):
        device.plan_order(np.array([0, 9], np.int32), 9)


def test_pageable_result_blocks_are_recycled_only_when_nobody_holds_them():
    """device._result_array (the pageable results of the host-widened copy): a block is laid under a new result only
    once the earlier result and every view of it -- numpy or torch -- are gone; a small result never takes a huge
    block; the cache forgets its oldest blocks beyond its budget."""
    import torch
    from prosstt_amd import device
    saved = list(device._result_blocks)
    del device._result_blocks[:]
    try:
        a = device._result_array((300, 1000), np.int64)
        b = device._result_array((300, 1000), np.int64)
        pa, pb = a.ctypes.data, b.ctypes.data
        assert pa != pb and a.flags.writeable and a.flags.c_contiguous and a.dtype == np.int64 and a.shape == (300, 1000)
        view = a[5:9]
        del a
        c = device._result_array((300, 1000), np.int64)
        assert c.ctypes.data not in (pa, pb)                     # the view still holds a's block
        pc = c.ctypes.data
        del view, c
        d = device._result_array((280, 1000), np.int32)          # smaller, another type: laid over a freed block
        assert d.ctypes.data in (pa, pc) and d.dtype == np.int32 and d.shape == (280, 1000)
        pd = d.ctypes.data
        t = torch.from_numpy(d)
        del d
        assert device._result_array((280, 1000), np.int32).ctypes.data != pd      # held through the tensor
        del t
        assert device._result_array((3, 3), np.int64).ctypes.data not in (pa, pb, pc)   # far smaller: its own block
        held = sum(x.size for x in device._result_blocks)
        assert held >= 2 * 300 * 1000 * 8
        old = device.RESULT_CACHE_BYTES
        device.RESULT_CACHE_BYTES = 1000
        try:
            device._result_array((3, 3), np.int64)
            assert len(device._result_blocks) == 1
        finally:
            device.RESULT_CACHE_BYTES = old
        assert np.array_equal(b, b)                              # (b is still the caller's: dropping it from the cache freed nothing)
        device._result_array((50, 50), np.int64)
        device.release_result_memory()
        assert device._result_blocks == []
    finally:
        device._result_blocks[:] = saved
