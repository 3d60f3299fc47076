"""CPU: the balance-beam oracle (oracle/balance_oracle.c) against one-step transitions produced by the
reference's own checker world (envs/balance_beam_env.py PantheonLine, through
tests/golden/make_balance_golden.py) -- observation rows of both agents, reward, done -- and the reset
stream against an independent restatement of rng.hpp + resetWorld (sim.cpp:45-74)."""
import os

import numpy as np

from conftest import GOLDEN


def test_oracle_reproduces_reference_transitions(oracle_lib):
    z = np.load(os.path.join(GOLDEN, "balance_transitions.npz"))
    before, acts, after, rew, done = z["before"], z["actions"], z["after"], z["reward"], z["done"]
    n = len(before)
    orc = oracle_lib.BalanceOracle(n)
    orc.plant(np.ascontiguousarray(before.transpose(1, 0, 2)))          # (n, 2, 7) -> (2, n, 7)
    orc.step(np.ascontiguousarray(acts.T))
    assert np.array_equal(orc.done, done)
    assert np.allclose(orc.reward[0], rew, rtol=0, atol=1e-7) and np.array_equal(orc.reward[0], orc.reward[1])
    alive = done == 0                                                   # the reference compares next states unless done
    assert np.array_equal(orc.obs[:, alive].transpose(1, 0, 2), after[alive])
    assert alive.sum() > 1500 and (~alive).sum() > 3000
    # a finished world is a pristine new episode: full clock, empty history, both views consistent
    fresh = orc.obs[:, ~alive]
    assert (fresh[:, :, 6] == 2).all() and (fresh[:, :, [1, 2, 4, 5]] == 0).all()
    assert np.array_equal(fresh[0, :, 0], fresh[1, :, 3]) and np.array_equal(fresh[0, :, 3], fresh[1, :, 0])
    assert fresh[:, :, 0].min() >= 2 and fresh[:, :, 0].max() <= 6


def _seed(idx):
    m = 0xFFFFFFFF
    v0, v1, s0 = idx & m, 0, 0
    for _ in range(8):
        s0 = (s0 + 0x9E3779B9) & m
        v0 = (v0 + (((((v1 << 4) & m) + 0xA341316C) & m) ^ ((v1 + s0) & m) ^ (((v1 >> 5) + 0xC8013EA4) & m))) & m
        v1 = (v1 + (((((v0 << 4) & m) + 0xAD90777D) & m) ^ ((v0 + s0) & m) ^ (((v0 >> 5) + 0x7E95761E) & m))) & m
    return v0


def _positions(episode):
    v, out = _seed(episode), []
    for _ in range(2):
        v = (1664525 * v + 1013904223) & 0xFFFFFFFF
        out.append(int(np.float32(5) * (np.float32(v & 0xFFFFFF) / np.float32(0x1000000))))
    return out


def test_reset_stream_and_episode_order(oracle_lib):
    """World w starts as episode w; worlds that finish in a step take the next indices in ascending world order."""
    n = 300
    orc = oracle_lib.BalanceOracle(n)
    for w in range(n):
        a, b = _positions(w)
        assert orc.obs[0, w].tolist() == [a + 2, 0, 0, b + 2, 0, 0, 2] and orc.obs[1, w].tolist() == [b + 2, 0, 0, a + 2, 0, 0, 2]
    rng = np.random.default_rng(1)
    nxt = n
    for _ in range(12):
        orc.step(rng.integers(0, 4, size=(2, n)).astype(np.int32))
        for w in np.flatnonzero(orc.done):
            a, b = _positions(nxt)
            assert orc.obs[0, w].tolist() == [a + 2, 0, 0, b + 2, 0, 0, 2], (w, nxt)
            nxt += 1
        assert orc.episodes == nxt
    assert nxt > 3 * n
