"""
-m gpu: seeded random shapes and parameter mixes against the scalar C model, count for count.  What the
fixed cases of test_gpu_sampler.py pick by hand is drawn here: ragged N and G (one cell, one gene, G % 4 != 0,
strips and tiles one short of / one past a boundary), padded output rows (ld_out > G), arbitrary global cell
ids, dense and sparse gamma-Poisson genes (list regions that overflow next to regions that do not), long
inversion walks, library-size factors over six orders of magnitude, degenerate genes; and the same cells presented in
another order (grouped by mean-tensor row, or shuffled) give the same counts.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng):
    edge = [1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1000]
    N = int(rng.choice(edge + [int(rng.integers(1, 1500))]))
    G = int(rng.choice(edge + [int(rng.integers(1, 1500))]))
    rows = int(rng.integers(1, 40))
    base = np.exp(rng.normal(rng.uniform(-1.0, 2.5), rng.uniform(0.3, 1.6), G))
    means = (np.exp(rng.normal(0.0, 0.5, (rows, G))) * base).astype(np.float32)
    kind = rng.integers(0, 5)
    if kind == 1:                                   # a dense block of gamma-Poisson genes somewhere
        lo = int(rng.integers(0, G))
        means[:, lo:lo + int(rng.integers(1, 300))] *= float(rng.choice([50.0, 200.0, 3000.0]))
    elif kind == 2:                                 # sparse gamma-Poisson genes
        means[:, rng.random(G) < rng.uniform(0.0, 0.08)] *= 200.0
    elif kind == 3:                                 # long inversion walks
        means = np.exp(rng.uniform(np.log(10), np.log(110), (rows, G))).astype(np.float32)
    roc = rng.integers(0, rows, N).astype(np.int32)
    sc = np.exp(rng.normal(0, rng.choice([0.1, 0.7, 3.0]), N))
    al = np.exp(rng.normal(np.log(0.2), np.log(1.5), G))
    be = np.exp(rng.normal(0.0, np.log(1.5), G)) + 1
    if kind == 3:
        al, be = rng.uniform(0.0, 0.15, G), rng.uniform(1.5, 6.0, G)
    if kind == 4:                                   # degenerate genes: Poisson limit, zero variance parameters
        al[rng.random(G) < 0.3] = 0.0
        be[rng.random(G) < 0.3] = 1.0 + 1e-8
        be[rng.random(G) < 0.1] = 1.0
    return means, roc, sc, al, be


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PROSSTT_FUZZ_CASES", "150"))))
def test_random_case_equals_the_model(seed):
    import torch
    from prosstt_amd import device
    from oracle import nb_model
    ctx = device.get_context()
    rng = np.random.default_rng(1000 + seed)
    means, roc, sc, al, be = _case(rng)
    N, G = len(roc), means.shape[1]
    draw_seed = int(rng.integers(0, 2 ** 63))
    mode = seed % 3
    if mode == 0:                                   # contiguous global ids from an offset
        offset = int(rng.integers(0, 2 ** 40))
        got = ctx.sample_counts(means, roc, sc, al, be, seed=draw_seed, cell_offset=offset, check_domain=False)
        want = nb_model.sample_counts(means, roc, sc, al, be, draw_seed, cell_offset=offset)
    elif mode == 1:                                 # arbitrary global ids (a shard of a larger plan)
        ids = np.sort(rng.choice(10 * N + 5, N, replace=False)).astype(np.int64)
        got = ctx.sample_counts(means, roc, sc, al, be, seed=draw_seed, cell_index=ids, check_domain=False)
        want = nb_model.sample_counts(means, roc, sc, al, be, draw_seed, cell_index=ids)
    else:                                           # rows of a wider output buffer (ld_out > G), offset 0
        pad = int(rng.choice([1, 3, 4, 60]))
        buf = torch.full((N, G + pad), -7, dtype=torch.int32, device=ctx.torch_device)
        got = ctx.sample_counts(means, roc, sc, al, be, seed=draw_seed, out=buf[:, :G], check_domain=False)
        assert bool((buf[:, G:] == -7).all()), "the padding of the output rows was written"
        want = nb_model.sample_counts(means, roc, sc, al, be, draw_seed)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    if seed % 2 == 0:
        # the same cells PRESENTED in another order (grouped by mean-tensor row as simulation.draw_counts presents them, or
        # shuffled), each keyed by its own global id: row i of the result is the cell presented i-th, count for count
        ids = (np.arange(N, dtype=np.int64) + offset) if mode == 0 else (ids if mode == 1 else np.arange(N, dtype=np.int64))
        order = device.plan_order(roc, means.shape[0]) if seed % 4 == 0 else rng.permutation(N).astype(np.int32)
        again = ctx.sample_counts(means, roc[order], sc[order], al, be, seed=draw_seed, cell_index=ids[order], check_domain=False)
        np.testing.assert_array_equal(again.cpu().numpy(), want[order])


@pytest.mark.parametrize("seed", range(40))
def test_random_lineage_attempt_batches_match_numpy(seed):
    """K2a over random shapes: B attempts x T steps x K programs against G genes and 0..3 siblings of other
    lengths -- the maximum of programs @ H and the per-sibling counts of anticorrelated genes (simulation.py:269-272,
    sim_utils.py:145-168) against binary64 numpy; a count may differ by the genes whose r is 0 to rounding."""
    import torch
    from prosstt_amd import device
    from oracle import ref_numpy
    ctx = device.get_context()
    rng = np.random.default_rng(7000 + seed)
    B, T, K = int(rng.integers(1, 20)), int(rng.integers(2, 70)), int(rng.integers(1, 45))
    G = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, int(rng.integers(1, 4000))]))
    walk = lambda t: rng.normal(0, 0.1, (t, K)).cumsum(axis=0) + np.log(rng.uniform(0.05, 1.5, K))
    P = np.stack([walk(T) for _ in range(B)])
    H = rng.standard_gamma(0.05, (K, G))
    if G > 3:
        H[:, G // 3] = 0.0                           # a constant gene: r is NaN, never < 0
    sibs = [walk(int(rng.integers(2, 80))) for _ in range(int(rng.integers(0, 4)))]
    tops, counts = ctx.lineage_attempt_batch(P, torch.as_tensor(H, device=ctx.torch_device), sibs)
    assert tops.shape == (B,) and counts.shape == (B, len(sibs))
    for b in range(B):
        rel = P[b] @ H
        assert tops[b] == pytest.approx(rel.max(), rel=1e-12, abs=1e-12)
        for j, S in enumerate(sibs):
            c = min(T, S.shape[0])
            r = ref_numpy.pearson_columns(rel[:c], (S @ H)[:c])
            with np.errstate(invalid="ignore"):
                want, ties = int(np.sum(r < 0)), int(np.sum(np.abs(r) < 1e-12))
            assert abs(int(counts[b, j]) - want) <= ties
