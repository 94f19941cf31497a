"""
-m gpu: the drop-in Python surface (prosstt_amd.tree / simulation / sim_utils /
count_model) end to end on the device, against the golden vectors of the real
reference and against the oracle.

Parity classes (SURVEY section 8 a/c):
  programs, coefficients, attempt counts, (pseudotime, branch, scaling): bit-exact;
  relative means (binary64 on device): rtol 1e-12;
  mean tensor (binary32 storage): rtol 1.2e-7 = one binary32 rounding;
  counts: bit-exact against the C model of the sampler on the device's own means,
          and first-moment agreement with the reference's draw.
"""
import json

import numpy as np
import pytest

from conftest import load_golden, tree_spec

pytestmark = pytest.mark.gpu

TREE_NAMES = ["bifurcation", "chain6", "chainbif7", "star5", "unequal"]


def make_tree(spec):
    from prosstt_amd import tree as ptree
    return ptree.Tree(topology=spec["topology"], time=spec["time"], num_branches=len(spec["time"]),
                      branch_points=spec["branch_points"], modules=spec["modules"], G=spec["G"])


@pytest.mark.parametrize("name,mode", [(n, "gamma") for n in TREE_NAMES] +
                         [("bifurcation", "beta"), ("star5", "beta")])
def test_simulate_lineage_vs_reference(name, mode):
    from prosstt_amd import simulation as sim
    g = load_golden("g3_lineage_%s_%s" % (name, mode))
    spec = tree_spec(g)
    np.random.seed(spec["seed"])
    t = make_tree(spec)
    stats = []
    rel, prog, H = sim.simulate_lineage(t, intra_branch_tol=0, stats=stats, **spec["kwargs"])
    assert np.random.random_sample() == float(g["state_after"])      # same stream consumption
    np.testing.assert_array_equal(H, g["H"])
    assert len(stats) == int(g["attempts"])
    assert [str(b) for b in rel.index] == list(g["bfs"])
    for b in t.branches:
        np.testing.assert_array_equal(prog[b], g["prog_%s" % b])
        np.testing.assert_allclose(rel[b], g["rel_%s" % b], rtol=1e-12, atol=1e-13)
        assert rel[b].dtype == np.float64 and rel[b].shape == (t.time[b], t.G)
    # per-attempt anticorrelated-gene counts: the reference evaluates every pair of present
    # siblings; the pairs that contain the branch under test are the ones that decide
    ref_counts = [json.loads(s) for s in g["anticorr"]]
    parallel = t.get_parallel_branches()
    accepted = []
    for idx, ((branch, top, counts), ref) in enumerate(zip(stats, ref_counts)):
        sibs = next((list(v) for v in parallel.values() if branch in v), [branch])
        present = sorted(s for s in sibs if s == branch or any(s == d for d in accepted))
        pairs = [(i, j) for i in range(len(present) - 1) for j in range(i + 1, len(present))]
        me = present.index(branch)
        assert counts == [ref[k] for k, (i, j) in enumerate(pairs) if me in (i, j)]
        if idx + 1 == len(stats) or stats[idx + 1][0] != branch:
            accepted.append(branch)


@pytest.mark.parametrize("name", ["bifurcation", "unequal", "star5"])
def test_full_pipeline_vs_reference(name):
    from prosstt_amd import simulation as sim, sim_utils as sut
    from oracle import nb_model
    g = load_golden("g6_sampling_%s" % name)
    spec = tree_spec(g)
    seed = spec["seed"]
    np.random.seed(seed)
    t = make_tree(spec)
    rel, prog, H = sim.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    base = sut.simulate_base_gene_exp(t, rel)
    np.testing.assert_array_equal(base, g["base"])            # same draws, same accept decisions
    t.add_genes(rel, base)
    for b in t.branches:
        np.testing.assert_allclose(t.means[b], g["means_%s" % b], rtol=1.2e-7, atol=1.2e-38)
    alpha = np.exp(np.random.normal(np.log(0.2), np.log(1.5), spec["G"]))
    beta = np.exp(np.random.normal(np.log(1), np.log(1.5), spec["G"])) + 1
    np.testing.assert_array_equal(alpha, g["alpha"])

    np.random.seed(seed + 1)
    X, pt, br, sc = sim.sample_density(t, len(g["pt"]), alpha=alpha, beta=beta, seed=1234)
    np.testing.assert_array_equal(pt, g["pt"])
    assert [str(b) for b in br] == list(g["br"])
    np.testing.assert_array_equal(sc, g["scalings"])
    assert X.dtype == np.int64 and X.shape == g["X"].shape
    # counts: bit-exact against the C model fed with the device's own mean tensor
    rows = sim.cell_rows(t, pt, br)
    want = nb_model.sample_counts(t.device_means().cpu().numpy(), rows, sc, alpha, beta, 1234)
    np.testing.assert_array_equal(X, want)
    # and statistically indistinguishable from the reference's draw in the first moment
    mu = g["mu"]
    var = (alpha * mu ** 2 + beta * mu).sum()
    assert abs(X.sum() - mu.sum()) < 6 * np.sqrt(var)
    assert abs(g["X"].sum() - mu.sum()) < 6 * np.sqrt(var)
    # deterministic intermediates against the reference's float64 (mu, p, r)
    from prosstt_amd import device
    dmu, dp, dr, _ = device.get_context().nb_params(t.device_means(), rows, sc, alpha, beta)
    np.testing.assert_allclose(dmu.cpu().numpy(), mu, rtol=1e-6)
    np.testing.assert_allclose(dp.cpu().numpy(), g["p"], rtol=1e-6)
    np.testing.assert_allclose(dr.cpu().numpy(), g["r"], rtol=2e-6)

    # the other samplers share the plan logic (bit-exact) and the kernel
    np.random.seed(seed + 2)
    X2, pt2, br2, sc2 = sim.sample_whole_tree(t, 2, alpha=alpha, beta=beta)
    np.testing.assert_array_equal(np.array(pt2), g["wt_pt"])
    np.testing.assert_array_equal(sc2, g["wt_sc"])
    assert X2.shape == g["wt_X"].shape
    np.random.seed(seed + 3)
    X3, pt3, br3, sc3 = sim._sample_data_at_times(t, np.arange(0, t.get_max_time(), 3), alpha=alpha, beta=beta)
    assert [str(b) for b in br3] == list(g["at_br"])
    np.testing.assert_array_equal(sc3, g["at_sc"])
    np.random.seed(seed + 4)
    X4, pt4, br4, sc4 = sim.sample_pseudotime_series(t, 30, [5, 30, 60], 6.0, alpha=0.3, beta=2)
    np.testing.assert_array_equal(pt4, g["ps_pt"])
    assert [str(b) for b in br4] == list(g["ps_br"]) and X4.shape == g["ps_X"].shape
    gp = dict(alpha=alpha[:7], beta=beta[:7], base_expr=base[:7])
    nd = sim.add_non_diff_genes(X, 7, gp, sc)
    assert nd.dtype == np.float64 and nd.shape == g["nd_X"].shape
    np.testing.assert_array_equal(nd[:, :X.shape[1]], X)


def test_config1_chain_200x500():
    """BASELINE.json configs[0]: 5-branch chain x 40 steps, sample_whole_tree(t, 1) -> 200 x 500."""
    from prosstt_amd import simulation as sim, sim_utils as sut, count_model as cm
    g = load_golden("g8_config1_chain")
    spec = tree_spec(g)
    np.random.seed(92)
    t = make_tree(spec)
    rel, _, H = sim.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    np.testing.assert_array_equal(H, g["H"])
    np.testing.assert_allclose(rel["E"], g["rel_E"], rtol=1e-12, atol=1e-13)
    base = sut.simulate_base_gene_exp(t, rel)
    np.testing.assert_array_equal(base, g["base"])
    t.add_genes(rel, base)
    alpha, beta = cm.generate_negbin_params(t)
    np.testing.assert_array_equal(alpha, g["alpha"])
    np.random.seed(93)
    X, pt, br, sc = sim.sample_whole_tree(t, 1, alpha=alpha, beta=beta)
    assert X.shape == (200, 500) and X.dtype == np.int64
    np.testing.assert_array_equal(np.array(pt), g["pt"])
    np.testing.assert_array_equal(sc, g["scalings"])
    # same law as the reference's matrix: totals agree within sampling error
    ref_total = int(g["total"])
    assert abs(X.sum() - ref_total) < 0.05 * ref_total
    assert abs((X == 0).mean() - (g["X"] == 0).mean()) < 0.02


def test_minimal_example_and_alias():
    """minimal_example.ipynb through the aliased import path; 4 return values."""
    import sys
    import prosstt_amd
    saved = {k: v for k, v in sys.modules.items() if k == "prosstt" or k.startswith("prosstt.")}
    try:
        prosstt_amd.install_as_prosstt()
        from prosstt import tree, simulation as sim
        np.random.seed(92)
        t = tree.Tree()
        out = sim.sample_whole_tree_restricted(t)
        assert len(out) == 4
        X, pt, br, sc = out
        g = load_golden("g8_minimal_example")
        assert X.shape == (80, 500)
        np.testing.assert_array_equal(pt, g["pt"])
        assert [str(b) for b in br] == list(g["br"])
        np.testing.assert_array_equal(sc, g["scalings"])
        assert abs(X.sum() - int(g["total"])) < 0.1 * int(g["total"])
        import torch
        Xd, cell_of_row = sim.draw_counts(t, pt, br, sc, [0.2] * t.G, [3.0] * t.G, seed=5, out="torch")
        assert Xd.dtype == torch.int32 and Xd.is_cuda and sorted(cell_of_row.tolist()) == list(range(80))
        with pytest.raises(ValueError):       # exact-zero mean -> scipy's domain error in the reference
            t.means = {b: np.zeros((40, 500)) for b in t.branches}
            sim.draw_counts(t, pt, br, sc, [0.2] * t.G, [3.0] * t.G)
    finally:
        for k in [k for k in sys.modules if k == "prosstt" or k.startswith("prosstt.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_sample_counts_new_name():
    from prosstt_amd import count_model as cm
    from oracle import nb_model
    rng = np.random.default_rng(0)
    mu = np.exp(rng.normal(0.5, 1.2, (50, 64)))
    X = cm.sample_counts(mu, 0.2, 2.0, seed=9)
    want = nb_model.sample_counts(mu.astype(np.float32), np.arange(50), np.ones(50), np.full(64, 0.2),
                                  np.full(64, 2.0), 9)
    np.testing.assert_array_equal(X, want)
    assert X.dtype == np.int64


def test_compact_host_returns_and_chunk_generator():
    """out="numpy32" / "numpy16": the same counts as the reference-typed int64 return in a half / a quarter of the
    bytes (uint16 refuses a count it cannot hold); sample_density_chunks: the chunks of one plan concatenate to the
    matrix of the single call, whatever the chunk length, and a zero mean raises from the generator as it does
    from the call."""
    from prosstt_amd import count_model as cm, simulation as sim
    from prosstt_amd import tree as ptree
    np.random.seed(7)
    t = ptree.Tree(topology=[["A", "B"], ["A", "C"]], time={"A": 30, "B": 25, "C": 35}, num_branches=3, branch_points=1,
                   modules=6, G=700)
    rel, _, _ = sim.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    from prosstt_amd import sim_utils as sut
    t.add_genes(rel, sut.simulate_base_gene_exp(t, rel))
    al, be = np.full(t.G, 0.3), np.full(t.G, 2.0)
    state = np.random.get_state()
    X, pt, br, sc = sim.sample_density(t, 1000, alpha=al, beta=be, seed=11)
    assert X.dtype == np.int64
    for out, dtype in (("numpy32", np.int32), ("numpy16", np.uint16)):
        np.random.set_state(state)
        Y = sim.sample_density(t, 1000, alpha=al, beta=be, seed=11, out=out)[0]
        assert Y.dtype == dtype and np.array_equal(Y, X)
    # out="csr": the same matrix, compacted on the device (rows in plan order, columns ascending within a row)
    import scipy.sparse as sparse
    np.random.set_state(state)
    S = sim.sample_density(t, 1000, alpha=al, beta=be, seed=11, out="csr")[0]
    assert sparse.isspmatrix_csr(S) and S.dtype == np.int32 and S.shape == X.shape and S.indices.dtype == np.int32
    assert S.nnz == np.count_nonzero(X) and np.all(S.data != 0)
    assert np.array_equal(S.toarray(), X)
    assert all(np.all(np.diff(S.indices[a:b]) > 0) for a, b in zip(S.indptr[:-1], S.indptr[1:]))
    from prosstt_amd import device
    np.random.set_state(state)
    presented = sim.sample_density(t, 1000, alpha=al, beta=be, seed=11, out="torch")[0]
    for chunk_bytes in (4 * t.G * 7, 4 * t.G * 1000, 1 << 28):        # many chunks (a ragged last one), one chunk exactly, one
        for row_order in (presented.cell_of_row, None):
            S = device.to_host_csr(presented.counts, chunk_bytes=chunk_bytes, row_order=row_order)
            want = X if row_order is not None else X[presented.cell_of_row]
            assert np.array_equal(S.toarray(), want) and S.has_sorted_indices
    assert np.array_equal(presented.to_host("csr").toarray(), X)
    empty = device.to_host_csr(presented.counts[:0])
    assert empty.shape == (0, t.G) and empty.nnz == 0
    zeros = device.to_host_csr(presented.counts[:5] * 0)
    assert zeros.shape == (5, t.G) and zeros.nnz == 0 and np.array_equal(zeros.indptr, np.zeros(6))
    for chunk in (1000, 333, 64, 5000):
        np.random.set_state(state)
        parts = list(sim.sample_density_chunks(t, 1000, chunk, alpha=al, beta=be, seed=11, out="numpy16"))
        assert len(parts) == -(-1000 // chunk) and all(p[0].dtype == np.uint16 for p in parts)
        assert np.array_equal(np.concatenate([p[0] for p in parts]), X)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), pt)
        assert list(np.concatenate([p[2] for p in parts])) == list(br)
        assert np.array_equal(np.concatenate([p[3] for p in parts]), sc)
    np.random.set_state(state)
    parts = list(sim.sample_density_chunks(t, 1000, 333, alpha=al, beta=be, seed=11, out="csr"))
    assert np.array_equal(sparse.vstack([p[0] for p in parts]).toarray(), X)
    assert list(sim.sample_density_chunks(t, 0, 10, alpha=al, beta=be, seed=1)) == []
    # a count beyond 65 535 does not fit uint16
    mu = np.full((4, 8), 3.0e5)
    assert cm.sample_counts(mu, 0.01, 1.5, seed=3, out="numpy32").max() > 65535
    with pytest.raises(OverflowError):
        cm.sample_counts(mu, 0.01, 1.5, seed=3, out="numpy16")
    # the domain check travels with the chunks
    bad = {b: np.array(t.means[b]) for b in t.branches}
    bad["B"][3, 5] = 0.0
    t.means = bad
    with pytest.raises(ValueError):
        for _ in sim.sample_density_chunks(t, 2000, 500, alpha=al, beta=be, seed=11):
            pass


def test_host_returns_widened_on_the_host_equal_the_device_widened_ones(monkeypatch):
    """int64 / int32 returns of 2^24 counts and more cross PCIe in a wire format (the low 8 bits of every count and, beside
    them, the entries that have higher bits set; the low 16 bits, then int32, when there are too many of those) and are
    widened by the host library's threads under the next chunk's transfer == widened on the device
    (PROSSTT_AMD_WIDEN=device) == the device tensor: with and without the row gather, for one chunk, many chunks and a
    ragged last chunk, for matrices that fit each wire with a few exceptions (counts above it, negative entries) and one
    that fits none."""
    import torch
    from prosstt_amd import device
    gen = torch.Generator(device="cuda").manual_seed(5)
    n, g = 4100, 4100                                   # 1.68e7 >= 2^24
    order = np.random.default_rng(1).permutation(n)

    def with_exceptions(base, row_add):
        m = base.clone()
        m[n // 2, 7], m[0, 0], m[n - 1, g - 2], m[3, 3], m[1000, 4000] = 65536, 84036, 2 ** 31 - 1, -1, -2 ** 31
        m[17, 5], m[n - 1, g - 1] = 65535, 0
        m[2000] += row_add                              # a whole row above the wire (under one entry in 256 of any chunk here)
        return m

    small = with_exceptions(torch.randint(0, 256, (n, g), device="cuda", generator=gen, dtype=torch.int32), 300)
    medium = with_exceptions(torch.randint(0, 65536, (n, g), device="cuda", generator=gen, dtype=torch.int32), 70000)
    large = torch.randint(0, 65536, (n, g), device="cuda", generator=gen, dtype=torch.int32) + 100000
    calls = []
    real = device._to_host_widened
    monkeypatch.setattr(device, "_to_host_widened", lambda *a: (calls.append(a[4]), real(*a))[1])
    for name, counts, wires in (("small", small, ["u8"]), ("medium", medium, ["u8", "u16"]), ("large", large, ["u8", "u16", "i32"])):
        want32 = counts.cpu().numpy()
        for dtype in (np.int64, np.int32):
            want = want32.astype(dtype)
            want_perm = np.empty_like(want)
            want_perm[order] = want
            for where in ("host", "device"):
                monkeypatch.setattr(device, "WIDEN_ON", where)
                for chunk_bytes in (256 << 20, 4 * g * 1000, 4 * g * 333):
                    del calls[:]
                    got = device.to_host(counts, dtype, chunk_bytes=chunk_bytes)
                    assert got.dtype == dtype and np.array_equal(got, want), (name, dtype, where, chunk_bytes)
                    if where == "host":                 # (int32 of a matrix no narrow wire can carry: copied as it lies)
                        assert calls == [w for w in wires if not (w == "i32" and dtype == np.int32)], (name, dtype, calls)
                    got = device.to_host(counts, dtype, chunk_bytes=chunk_bytes, row_order=order)
                    assert np.array_equal(got, want_perm), (name, dtype, where, chunk_bytes)
    monkeypatch.setattr(device, "WIDEN_ON", "host")
    for narrowest, tried in (("i32", ["i32"]), ("u16", ["u16"])):
        monkeypatch.setattr(device, "WIRE", narrowest)
        del calls[:]
        assert np.array_equal(device.to_host(medium, np.int64, chunk_bytes=4 * g * 777), medium.cpu().numpy().astype(np.int64)) and calls == tried
    monkeypatch.setattr(device, "WIRE", "u8")
    monkeypatch.setattr(device, "HOST_THREADS", 3)
    assert np.array_equal(device.to_host(small, np.int64, chunk_bytes=4 * g * 777, row_order=order)[order], small.cpu().numpy())
    # the result in ordinary memory (a numpy array; PROSSTT_AMD_RESULT_MEMORY=pageable) instead of page-locked memory
    monkeypatch.setattr(device, "RESULT_MEMORY", "pageable")
    for dtype in (np.int64, np.int32):
        got = device.to_host(small, dtype, chunk_bytes=4 * g * 500, row_order=order)
        assert got.dtype == dtype and got.flags.writeable and np.array_equal(got[order], small.cpu().numpy())


def test_sparse_return_over_the_narrow_wire(monkeypatch):
    """to_host_csr of a matrix with 2^22 non-zeros and more: values as their low 8 bits (+ the larger ones as pairs), column
    indices as 16 bits, widened by the host library == the dense matrix; a matrix of large counts falls back to 4-byte
    values; WIRE / WIDEN_ON switch the narrowing off; rows in plan order through row_order."""
    import scipy.sparse as sparse
    import torch
    from prosstt_amd import device
    gen = torch.Generator(device="cuda").manual_seed(9)
    n, g = 4100, 4100
    keep = torch.rand((n, g), device="cuda", generator=gen) < 0.4
    small = torch.randint(1, 256, (n, g), device="cuda", generator=gen, dtype=torch.int32) * keep
    small[5, 5], small[n - 1, g - 1], small[100, 0], small[7] = 84036, 256, 2 ** 31 - 1, 0
    small[9, :64] = 70000
    large = (torch.randint(300, 100000, (n, g), device="cuda", generator=gen, dtype=torch.int32)) * keep
    order = np.random.default_rng(2).permutation(n)
    for name, counts in (("small", small), ("large", large)):
        want = counts.cpu().numpy()
        assert np.count_nonzero(want) >= 1 << 22
        want_perm = np.empty_like(want)
        want_perm[order] = want
        for wire, where in (("u8", "host"), ("u16", "host"), ("i32", "host"), ("u8", "device")):
            monkeypatch.setattr(device, "WIRE", wire)
            monkeypatch.setattr(device, "WIDEN_ON", where)
            for chunk_bytes in (256 << 20, 4 * g * 1000, 4 * g * 333):
                S = device.to_host_csr(counts, chunk_bytes=chunk_bytes)
                assert sparse.isspmatrix_csr(S) and S.dtype == np.int32 and S.indices.dtype == np.int32 and S.has_sorted_indices
                assert S.nnz == np.count_nonzero(want) and np.array_equal(S.toarray(), want), (name, wire, where, chunk_bytes)
            S = device.to_host_csr(counts, chunk_bytes=4 * g * 777, row_order=order)
            assert np.array_equal(S.toarray(), want_perm), (name, wire, where)


def test_max_attempts_guard():
    from prosstt_amd import simulation as sim
    from prosstt_amd import tree as ptree
    np.random.seed(1)
    t = ptree.Tree(G=2000, modules=10)
    with pytest.raises(RuntimeError):
        sim.simulate_lineage(t, rel_exp_cutoff=-50, a=0.05, max_attempts=3)


def test_secondary_samplers_counts_bit_exact_vs_model():
    """sample_whole_tree, sample_pseudotime_series, _sample_data_at_times and add_non_diff_genes
    (simulation.py:474-517, 319-379, 551-599, 654-675) with an explicit sampler seed: every count --
    the appended non-differential columns included -- equals the C model's for the plan the call
    returned."""
    from prosstt_amd import simulation as sim, sim_utils as sut
    from oracle import nb_model
    np.random.seed(31)
    t = make_tree(dict(topology=[["A", "B"], ["A", "C"], ["C", "D"], ["C", "E"]],
                       time={"A": 30, "B": 45, "C": 25, "D": 20, "E": 33}, branch_points=2, G=300, modules=6,
                       int_labels=False))
    rel, _, _ = sim.simulate_lineage(t, a=0.05, intra_branch_tol=0)
    base = sut.simulate_base_gene_exp(t, rel)
    t.add_genes(rel, base)
    G = t.G
    alpha = np.exp(np.random.normal(np.log(0.2), np.log(1.5), G))
    beta = np.exp(np.random.normal(0, np.log(1.5), G)) + 1
    host_means = t.device_means().cpu().numpy()

    def model(pt, br, sc, al, be, seed):
        rows = sim.cell_rows(t, np.asarray(pt), np.asarray(br))
        return nb_model.sample_counts(host_means, rows, np.asarray(sc), np.broadcast_to(al, (G,)).astype(float),
                                      np.broadcast_to(be, (G,)).astype(float), seed)

    X, pt, br, sc = sim.sample_whole_tree(t, 2, alpha=alpha, beta=beta, seed=101)
    assert X.dtype == np.int64 and X.shape == (len(pt), G)
    np.testing.assert_array_equal(X, model(pt, br, sc, alpha, beta, 101))
    X, pt, br, sc = sim.sample_pseudotime_series(t, 60, [5, 40, 70], 5.0, alpha=0.3, beta=2, seed=102)
    np.testing.assert_array_equal(X, model(pt, br, sc, 0.3, 2.0, 102))
    X, pt, br, sc = sim._sample_data_at_times(t, np.arange(0, t.get_max_time(), 2), alpha=alpha, beta=beta, seed=103)
    np.testing.assert_array_equal(X, model(pt, br, sc, alpha, beta, 103))
    X, pt, br, sc = sim.sample_density(t, 150, alpha=alpha, beta=beta, seed=104)
    np.testing.assert_array_equal(X, model(pt, br, sc, alpha, beta, 104))
    # add_non_diff_genes: one constant mean row, the cells' own scalings, its own seed
    extra = 9
    gp = dict(alpha=alpha[:extra], beta=beta[:extra], base_expr=base[:extra])
    wide = sim.add_non_diff_genes(X, extra, gp, sc, seed=105)
    assert wide.dtype == np.float64 and wide.shape == (150, G + extra)
    np.testing.assert_array_equal(wide[:, :G], X)
    want = nb_model.sample_counts(base[:extra].astype(np.float32).reshape(1, extra), np.zeros(150, np.int32),
                                  sc, alpha[:extra], beta[:extra], 105)
    np.testing.assert_array_equal(wide[:, G:], want)


def test_device_returns_present_cells_grouped_and_carry_the_permutation():
    """out="torch": the default return is the matrix as it lies on the device -- cells presented grouped by their row of the
    mean tensor (the regime bench.py times) -- with the permutation; order="plan" presents the cells as planned.  Both hold
    the same counts, bit for bit: grouped return + permutation == plan-order return == the host array; the chunked
    generator yields the same, chunk by chunk."""
    import torch
    from prosstt_amd import device, simulation as sim, workloads
    work = workloads.build("C2", G=1024)
    t, al, be = work.tree, work.alpha, work.beta
    np.random.seed(21)
    planned, pt, br, sc = sim.sample_density(t, 1500, alpha=al, beta=be, seed=17, out="torch", order="plan")
    np.random.seed(21)
    presented, pt2, br2, sc2 = sim.sample_density(t, 1500, alpha=al, beta=be, seed=17, out="torch")
    np.random.seed(21)
    host = sim.sample_density(t, 1500, alpha=al, beta=be, seed=17, out="numpy32")[0]
    assert isinstance(presented, device.PresentedCounts) and torch.is_tensor(planned)
    np.testing.assert_array_equal(pt, pt2)
    counts, cell_of_row = presented
    assert counts.shape == (1500, 1024) == presented.shape and sorted(cell_of_row.tolist()) == list(range(1500))
    rows = sim.cell_rows(t, pt, br)
    assert (np.diff(rows[cell_of_row]) >= 0).all() and (np.diff(rows) < 0).any()      # grouped by row; the plan is not
    assert torch.equal(counts, planned[torch.as_tensor(cell_of_row, device=planned.device)])
    assert torch.equal(presented.in_plan_order(), planned)
    np.testing.assert_array_equal(presented.to_host("numpy32"), host)
    np.testing.assert_array_equal(planned.cpu().numpy(), host)
    with pytest.raises(ValueError):
        sim.sample_density(t, 10, alpha=al, beta=be, seed=17, out="torch", order="sorted")
    for order in ("presented", "plan"):
        np.random.seed(21)
        got = np.empty_like(host)
        lo = 0
        for part, ppt, pbr, psc in sim.sample_density_chunks(t, 1500, 400, alpha=al, beta=be, seed=17, out="torch", order=order):
            hi = lo + len(ppt)
            if order == "presented":
                c, perm = part
                got[lo + perm] = c.cpu().numpy()
            else:
                got[lo:hi] = part.cpu().numpy()
            lo = hi
        np.testing.assert_array_equal(got, host)
