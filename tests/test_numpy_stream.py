"""
CPU tests of the library's generator of numpy's legacy global stream (prosstt_amd/csrc/numpy_stream.h through
prosstt_amd_numpy_programs; no GPU involved): a batch of candidate expression programs drawn in one call equals, bit for
bit, the reference's one-walk-at-a-time draws -- values, the state behind every attempt, the position numpy is left at.
The reference side here is numpy itself, called the way simulation.diffusion calls it (simulation.py:104-121); the
programs of the real reference are pinned by fixtures g2/g3 in test_gpu_pipeline.py / test_oracle_golden.py.
"""
import numpy as np
import pytest

from prosstt_amd import simulation as sim


def _reference_walk(steps):
    """simulation.diffusion as the reference writes it (simulation.py:104-121)."""
    start = np.random.uniform(0, 1.5)
    vel0 = np.random.normal(0, 0.2)
    eta = np.random.uniform(0, 1)
    eps = np.random.normal(0, 2 / steps, steps - 1) if steps > 1 else np.zeros(0)
    walk, velocity = np.zeros(steps), np.zeros(steps)
    walk[0], velocity[0] = np.log(start), vel0
    for t in range(steps - 1):
        walk[t + 1] = walk[t] + velocity[t]
        velocity[t + 1] = eta * velocity[t] + eps[t]
    return walk


@pytest.mark.parametrize("seed,attempts,T,K", [(0, 1, 50, 25), (92, 7, 50, 25), (3, 16, 40, 6), (12345, 5, 1, 3),
                                               (7, 3, 2, 2), (2 ** 31, 9, 33, 24)])
def test_batch_equals_sequential_numpy_calls(seed, attempts, T, K):
    np.random.seed(seed)
    np.random.standard_normal(3)               # leave a cached Gaussian behind (odd number of normals)
    np.random.random_sample(17)
    entry = np.random.get_state()
    want, want_states = [], []
    for _ in range(attempts):
        want.append(np.transpose([_reference_walk(T) for _ in range(K)]))
        want_states.append(np.random.get_state())
    tail_want = np.random.random_sample(4)
    np.random.set_state(entry)
    got, states = sim.sim_expr_branches(attempts, T, K)
    assert got.shape == (attempts, T, K)
    for a in range(attempts):
        assert np.array_equal(got[a], want[a]), "attempt %d differs" % a
        for x, y in zip(states[a][1:], want_states[a][1:]):
            assert np.array_equal(x, y)
    assert np.array_equal(np.random.random_sample(4), tail_want)          # numpy is left behind the last attempt
    # rewinding to an accepted attempt: the stream goes on from there
    np.random.set_state(states[0])
    np.random.set_state(want_states[0])
    a = np.random.standard_normal(5)
    np.random.set_state(states[0])
    assert np.array_equal(np.random.standard_normal(5), a)


def test_matches_the_single_attempt_function():
    np.random.seed(4)
    one = sim.sim_expr_branch(50, 25)
    np.random.seed(4)
    many, _ = sim.sim_expr_branches(1, 50, 25)
    assert np.array_equal(one, many[0])


def test_across_a_block_boundary_of_the_generator():
    """624 words per block: a batch that spans many regenerations, entered at every position class."""
    for burn in (0, 1, 311, 623, 624, 625):
        np.random.seed(11)
        np.random.randint(0, 2 ** 32, size=burn, dtype=np.uint32)
        entry = np.random.get_state()
        want = [np.transpose([_reference_walk(30) for _ in range(4)]) for _ in range(12)]
        end = np.random.get_state()
        np.random.set_state(entry)
        got, states = sim.sim_expr_branches(12, 30, 4)
        assert np.array_equal(got, np.stack(want))
        assert all(np.array_equal(x, y) for x, y in zip(np.random.get_state()[1:], end[1:]))


def test_refuses_a_single_program():
    with pytest.raises(ValueError):
        sim.sim_expr_branches(2, 10, 1)
