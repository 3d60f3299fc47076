"""CPU: the Simplecooked (overcooked2_env) oracle (oracle/simplecooked_oracle.c) against the golden vectors
produced by the reference's numpy twin of that world (envs/overcooked2_reimplement.py, through
tests/golden/make_simplecooked_golden.py).  This is what pins the oracle.

One byte per tomato-source cell is masked (`differs` in the fixture): the reference's C++ zeroes channel
5P+5 on every observation pass, wiping the TOMATO_SOURCE terrain bit that shares it
(src/overcooked2_env/sim.cpp:74,557); its numpy twin keeps the bit.  The oracle follows the C++."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "simplecooked_*.npz")))


def test_fixtures_present():
    assert len(FIXTURES) >= 11


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[13:-4])
def test_oracle_reproduces_reference(path, oracle_lib):
    z = np.load(path)
    params = json.loads(str(z["params"]))
    acts, obs, rew, done, keep = z["actions"], z["obs"], z["reward"], z["done"], ~z["differs"]
    n = 3
    orc = oracle_lib.SimplecookedOracle(params, n, num_threads=2)
    P = params["num_players"]
    assert orc.obs.shape == (n, P, params["height"] * params["width"], 5 * P + 10)
    assert np.array_equal(orc.obs[:, :, keep], np.broadcast_to(obs[0], orc.obs.shape)[:, :, keep])
    assert (orc.obs[:, :, ~keep] == 0).all()  # the C++ behaviour: the tomato-source bit is never visible
    total = 0
    for t in range(len(acts)):
        orc.step(np.repeat(acts[t].astype(np.int32)[:, None], n, axis=1))
        assert np.array_equal(orc.obs[:, :, keep], np.broadcast_to(obs[t + 1], orc.obs.shape)[:, :, keep]), f"obs, step {t}"
        assert (orc.reward == rew[t]).all() and orc.reward.shape == (P, n), f"reward, step {t}"
        assert (orc.done == done[t]).all(), f"done, step {t}"
        total += int(rew[t])
    if path.endswith("_cook.npz"):
        assert total > 100  # complete soup cycles, dish-pickup shaping included


def test_dishes_out_counts_dishes_on_counters(oracle_lib):
    """WorldState.num_dishes_out (sim.cpp:222-231) is redundant with the grid: always the number of
    dishes lying on counters.  The HIP engine keeps it as a per-world counter; this is the invariant."""
    from madrona_rl_envs_playground_amd import layouts
    params = layouts.get_simplecooked_layout_params("simple", 90)
    n = 600
    orc = oracle_lib.SimplecookedOracle(params, n, num_threads=4)
    rng = np.random.default_rng(5)
    terrain = np.array(params["terrain"])
    seen = 0
    for t in range(400):
        acts = rng.integers(0, 5, size=(2, n)).astype(np.int32)
        acts[rng.random((2, n)) < 0.5] = 5
        orc.step(acts)
        pl, ob, ts, dishes = orc.dump()
        on_counters = ((ob[:, :, 0] == 3) & (terrain == 2)[None, :]).sum(axis=1)
        assert np.array_equal(on_counters, dishes)
        seen = max(seen, int(dishes.max()))
    assert seen >= 2


def test_thread_count_does_not_change_results(oracle_lib):
    from madrona_rl_envs_playground_amd import layouts
    params = layouts.get_simplecooked_layout_params("random1", 40)
    n = 257
    a, b = oracle_lib.SimplecookedOracle(params, n, 1), oracle_lib.SimplecookedOracle(params, n, 8)
    rng = np.random.default_rng(3)
    for _ in range(100):
        acts = rng.integers(0, 6, size=(2, n)).astype(np.int32)
        a.step(acts)
        b.step(acts)
        assert np.array_equal(a.obs, b.obs) and np.array_equal(a.reward, b.reward) and np.array_equal(a.done, b.done)
