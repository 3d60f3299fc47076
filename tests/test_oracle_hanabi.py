"""CPU: the Hanabi oracle.  The reference holds no second implementation; what
its invariant checker tests (envs/hanabi_env.py:478-657) was run against this
oracle in the build container (tests/golden/make_hanabi_golden.py: 0 rejections
on the full configuration) and the accepted sequences are committed as fixtures.
Here: the oracle reproduces those fixtures, and the invariants are re-checked
independently of the reference's code.  Card-knowledge / last-action sections,
RNG draws and episode->seed mapping stay PARITY UNPINNED."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from madrona_rl_envs_playground_amd import hanabi_spec

CONFIGS = {
    "full": dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3),
    "small": dict(colors=2, ranks=5, players=2, max_information_tokens=3, max_life_tokens=1),
    "very_small": dict(colors=1, ranks=5, players=2, max_information_tokens=3, max_life_tokens=1),
}


def test_sizes():
    assert hanabi_spec.observation_size(CONFIGS["full"]) == 658      # OBS_SIZE, sim.hpp:29
    assert hanabi_spec.state_size(CONFIGS["full"]) == 783            # STATE_SIZE, sim.hpp:30
    assert hanabi_spec.num_moves(CONFIGS["full"]) == 20              # NUM_MOVES, sim.hpp:19
    assert hanabi_spec.num_moves(CONFIGS["small"]) == 17


@pytest.mark.parametrize("name", list(CONFIGS))
def test_oracle_reproduces_checked_fixture(name, oracle_lib):
    z = np.load(os.path.join(GOLDEN, f"hanabi_{name}.npz"))
    cfg = CONFIGS[name]
    n = z["actions"].shape[2]
    orc = oracle_lib.HanabiOracle(cfg, n)
    assert np.array_equal(orc.obs, z["first_obs"]) and np.array_equal(orc.state, z["first_state"])
    assert np.array_equal(orc.mask, z["first_mask"]) and np.array_equal(orc.active, z["first_active"])
    for t in range(z["actions"].shape[0]):
        orc.step(z["actions"][t].astype(np.int32))
        assert np.array_equal(orc.obs, z["obs"][t]), f"obs, step {t}"
        assert np.array_equal(orc.state, z["state"][t]), f"state, step {t}"
        assert np.array_equal(orc.mask, z["mask"][t]), f"mask, step {t}"
        assert np.array_equal(orc.active, z["active"][t]), f"active, step {t}"
        assert np.array_equal(orc.reward, z["reward"][t]), f"reward, step {t}"
        assert np.array_equal(orc.done, z["done"][t]), f"done, step {t}"


def decode(cfg, state_row):
    """Independent decoder of the sections the reference's checker looks at."""
    k, r = cfg["colors"], cfg["ranks"]
    bpc, at = k * r, 0
    partner = [state_row[at + c * bpc: at + (c + 1) * bpc] for c in range(5)]
    at += 5 * bpc
    short = state_row[at:at + 2]
    at += 2
    deck_bits = (4 + (r - 2) * 2) * k - 10
    deck = int(state_row[at:at + deck_bits].sum())
    assert state_row[at:at + deck].all() and not state_row[at + deck:at + deck_bits].any()
    at += deck_bits
    fireworks = []
    for c in range(k):
        seg = state_row[at:at + r]
        assert seg.sum() <= 1
        fireworks.append(0 if seg.sum() == 0 else int(seg.argmax()) + 1)
        at += r
    info = int(state_row[at:at + cfg["max_information_tokens"]].sum())
    at += cfg["max_information_tokens"]
    life = int(state_row[at:at + cfg["max_life_tokens"]].sum())
    at += cfg["max_life_tokens"]
    discards = []
    for c in range(k):
        for rank in range(r):
            copies = 3 if rank == 0 else (1 if rank == r - 1 else 2)
            seg = state_row[at:at + copies]
            discards.append(int(seg.sum()))
            assert seg[:discards[-1]].all()
            at += copies
    own_at = hanabi_spec.state_size(cfg) - 5 * bpc
    own = [state_row[own_at + c * bpc: own_at + (c + 1) * bpc] for c in range(5)]
    return dict(partner=partner, own=own, short=short, deck=deck, fireworks=fireworks, info=info, life=life,
                discards=discards)


@pytest.mark.parametrize("name", ["full", "small"])
def test_invariants(name, oracle_lib):
    cfg = CONFIGS[name]
    n, steps = 400, 120
    no, ns = hanabi_spec.observation_size(cfg), hanabi_spec.state_size(cfg)
    k, r = cfg["colors"], cfg["ranks"]
    deck_total = (4 + (r - 2) * 2) * k
    orc = oracle_lib.HanabiOracle(cfg, n, num_threads=4)
    rng = np.random.default_rng(1)
    prev_active = orc.active.copy()
    for t in range(steps):
        a = (rng.random(orc.mask.shape) * (orc.mask != 0)).argmax(-1).astype(np.int32)
        info_before = orc.dump()[:, 81].copy()
        orc.step(a)
        assert (orc.active.sum(0) == 1).all()                           # exactly one active agent
        flipped = orc.active[0] != prev_active[0]
        assert (flipped | (orc.done == 1)).all()                        # and it alternates unless the game ended
        assert (orc.active[0][orc.done == 1] == 1).all()                # a new game starts with agent 0
        assert (orc.reward[0] == orc.reward[1]).all()
        rec = orc.dump()
        for w in range(0, n, 9):
            cur = int(orc.active[1, w])
            if rec[w, 81] > cfg["max_information_tokens"] or info_before[w] > cfg["max_information_tokens"]:
                continue                                                # shifted encoding, see hanabi_oracle.c
            st = orc.state[cur, w, :ns]
            assert np.array_equal(st[:no], orc.obs[cur, w, :no])        # state prefix == obs
            d = decode(cfg, st)
            assert d["deck"] == rec[w, 50] and d["info"] == rec[w, 81] and d["life"] == rec[w, 82]
            hand_cards = sum(int(c.sum()) for c in d["partner"]) + sum(int(c.sum()) for c in d["own"])
            assert all(c.sum() <= 1 for c in d["partner"] + d["own"])
            assert hand_cards + d["deck"] + sum(d["fireworks"]) + sum(d["discards"]) == deck_total  # conservation
            if orc.done[w]:                                             # pristine start state after done
                assert d["deck"] == deck_total - 10 and sum(d["fireworks"]) == 0 and sum(d["discards"]) == 0
                assert d["info"] == cfg["max_information_tokens"] and d["life"] == cfg["max_life_tokens"]
                assert hand_cards == 10 and not d["short"].any()
        prev_active = orc.active.copy()


def test_stale_buffers_of_the_waiting_agent(oracle_lib):
    """sim.cpp:799-808: only the player to move is re-encoded."""
    cfg = CONFIGS["full"]
    n = 64
    orc = oracle_lib.HanabiOracle(cfg, n)
    rng = np.random.default_rng(2)
    for _ in range(30):
        before = orc.obs.copy()
        a = (rng.random(orc.mask.shape) * (orc.mask != 0)).argmax(-1).astype(np.int32)
        orc.step(a)
        for w in range(n):
            if not orc.done[w]:
                waiting = int(orc.active[0, w])  # index of the agent that is NOT active == 1 - argmax
                waiting = 0 if orc.active[1, w] else 1
                assert np.array_equal(orc.obs[waiting, w], before[waiting, w])
