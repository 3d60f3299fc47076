"""CPU: the Cartpole oracle against (a) one-step transitions computed by the
reference's float64 CartpoleNumpy (tests/golden/cartpole_transitions.npz,
tolerance 1e-6 as in envs/cartpole_env.py:277) and (b) known answers of the
episode-seeded generator worked out independently from rng.hpp:5-40."""
import os

import numpy as np

from conftest import GOLDEN

X_TH = 2.4
TH_TH = 12 * 2 * np.pi / 360


def test_one_step_matches_reference_float64(oracle_lib):
    z = np.load(os.path.join(GOLDEN, "cartpole_transitions.npz"))
    states, actions, next64, done = z["states"], z["actions"], z["next64"], z["done"]
    m = len(states)
    orc = oracle_lib.CartpoleOracle(m, num_threads=4)
    orc.state[:] = states
    orc.step(actions)
    # terminal flags may differ only when the float64 value sits within rounding of a threshold
    near = (np.abs(np.abs(next64[:, 0]) - X_TH) < 1e-6) | (np.abs(np.abs(next64[:, 2]) - TH_TH) < 1e-6)
    assert ((orc.done[:, 0] == done) | near).all()
    alive = (orc.done[:, 0] == 0) & (done == 0)
    assert alive.sum() > 3000
    assert np.abs(orc.state[alive].astype(np.float64) - next64[alive]).max() < 1e-6
    assert (orc.reward == 1).all()


def _seed(idx):
    """rng.hpp:7-26 restated with Python ints."""
    m = 0xFFFFFFFF
    v0, v1, s0 = idx & m, 0, 0
    for _ in range(8):
        s0 = (s0 + 0x9E3779B9) & m
        v0 = (v0 + ((((v1 << 4) & m) + 0xA341316C) & m ^ (v1 + s0) & m ^ ((v1 >> 5) + 0xC8013EA4) & m)) & m
        v1 = (v1 + ((((v0 << 4) & m) + 0xAD90777D) & m ^ (v0 + s0) & m ^ ((v0 >> 5) + 0x7E95761E) & m)) & m
    return v0


def _stream(idx, count):
    v, out = _seed(idx), []
    for _ in range(count):
        v = (1664525 * v + 1013904223) & 0xFFFFFFFF
        out.append(np.float32(v & 0xFFFFFF) / np.float32(0x1000000))
    return np.array(out, np.float32)


def test_generator_known_answers(oracle_lib):
    # literal values computed once with the independent restatement above
    assert _seed(0) == 4224205021 and _seed(1) == oracle_lib.rng_seed(1)
    for idx in (0, 1, 2, 3, 1023, 65535, 2 ** 31 + 5):
        assert oracle_lib.rng_seed(idx) == _seed(idx)
        assert np.array_equal(oracle_lib.rng_stream(idx, 6), _stream(idx, 6))
    first = oracle_lib.rng_stream(0, 3)
    assert np.allclose(first, [0.02851248, 0.16563553, 0.90761626], atol=1e-8)


def test_reset_states_and_episode_order(oracle_lib):
    n = 300
    orc = oracle_lib.CartpoleOracle(n)
    for w in (0, 1, 17, 299):
        r = _stream(w, 4)
        expect = np.float32(-0.05) + r * np.float32(0.1)
        assert np.array_equal(orc.state[w], expect.astype(np.float32))
    assert orc.episodes == n
    # drive every world over the edge: all reset, in ascending world order
    orc.state[:, 0] = 2.5
    orc.step(np.zeros(n, np.int32))
    assert orc.done.all() and orc.episodes == 2 * n
    for w in (0, 5, 299):
        expect = np.float32(-0.05) + _stream(n + w, 4) * np.float32(0.1)
        assert np.array_equal(orc.state[w], expect.astype(np.float32))
    # only some worlds reset: indices are handed out by rank among the resetting worlds
    orc.state[::3, 2] = 0.3
    orc.step(np.ones(n, np.int32))
    idx = np.flatnonzero(orc.done[:, 0])
    assert len(idx) == 100
    for k, w in enumerate(idx[:5]):
        expect = np.float32(-0.05) + _stream(2 * n + k, 4) * np.float32(0.1)
        assert np.array_equal(orc.state[w], expect.astype(np.float32))


def test_config0_plumbing_1024_worlds(oracle_lib):
    """BASELINE.json configs[0]: 1024 worlds, random actions, CPU."""
    n = 1024
    orc = oracle_lib.CartpoleOracle(n, num_threads=4)
    rng = np.random.default_rng(0)
    total_done = 0
    for _ in range(1000):
        orc.step(rng.integers(0, 2, n).astype(np.int32))
        total_done += int(orc.done.sum())
        assert np.isfinite(orc.state).all()
        assert (np.abs(orc.state[:, 0]) <= 2.4 + 1e-3).all()  # finished worlds were re-seeded
    assert 20000 < total_done < 80000  # random policy: episodes of ~20 steps
    assert orc.episodes == n + total_done
