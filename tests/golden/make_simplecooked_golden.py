"""Generates tests/golden/simplecooked_*.npz from the REFERENCE's own numpy implementation of the
overcooked2_env world (envs/overcooked2_reimplement.py, class DummyMDP), imported from
/root/reference in the build container.  Run only there:

    python tests/golden/make_simplecooked_golden.py              # write fixtures
    python tests/golden/make_simplecooked_golden.py --soak 20000   # extra lock-step check, nothing written

The fixtures hold data only: the simulator config, an action stream and the observations / rewards /
dones the reference produced for it.  Episode handling follows the reference's wrapper
(envs/overcooked2_env.py:288-305): done = timestep >= horizon, and on done the next observation is
that of a fresh start state.

It also cross-checks this repo's layout transform against the reference's
envs/overcooked2_env.py:get_base_layout_params (imported with stub modules for the uninstalled gym /
build.* packages and a file reader for the vendored old-style .layout files; ordinary
ModuleNotFoundError, no permission was denied).

One known difference between the reference's two implementations is kept OUT of the fixtures: the
numpy twin leaves the TOMATO_SOURCE terrain bit in channel 5P+5, the C++ zeroes that channel on every
observation pass (sim.cpp:74).  Layouts with tomato sources are therefore compared with that one
byte masked (see tests/test_oracle_simplecooked.py).
"""
import argparse
import ast
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(1, "/root/reference")

from madrona_rl_envs_playground_amd import layouts  # noqa: E402

from envs.overcooked2_reimplement import DummyMDP  # noqa: E402  (reference)

# fixture name -> (layout, horizon, steps, seed, mode); mode None = uniform, float = P(interact), "cook:<eps>" = goal-directed
CASES = {
    "simple": ("simple", 200, 700, 0, None),
    "simple_busy": ("simple", 60, 400, 1, 0.45),
    "unident_s": ("unident_s", 200, 500, 2, 0.35),
    "random1": ("random1", 200, 500, 3, 0.35),
    "random0": ("random0", 200, 500, 4, None),
    "random3": ("random3", 200, 500, 5, 0.35),
    "simple_tomato": ("simple_tomato", 120, 500, 6, 0.45),
    "simple_cook": ("simple", 200, 900, 9, "cook:0.1"),
    "random1_cook": ("random1", 250, 700, 10, "cook:0.15"),
    "unident_s_cook": ("unident_s", 200, 700, 11, "cook:0.1"),
    "simple_single": ("simple", 80, 300, 12, 0.4, 1),   # one player: no dish shaping (numpy: never useful)
}

DELTAS = {0: (0, -1), 1: (0, 1), 2: (1, 0), 3: (-1, 0)}  # action -> (dx, dy)
POT, COUNTER, ONION_SRC, DISH_SRC, SERVING, TOMATO_SRC = 1, 2, 3, 4, 5, 6


def sample_actions(rng, steps, players, p_interact):
    if p_interact is None:
        return rng.integers(0, 6, size=(steps, players)).astype(np.int8)
    a = rng.integers(0, 5, size=(steps, players))
    a[rng.random((steps, players)) < p_interact] = 5
    return a.astype(np.int8)


class Cook:
    """Goal-directed policy with epsilon noise (fetch ingredient -> pot (cooks by itself at three) -> dish
    -> plate -> serve; now and then park an item on a counter), so the fixtures contain complete soup
    cycles, dish-pickup shaping and dishes lying on counters."""

    def __init__(self, params, rng, eps):
        self.p, self.rng, self.eps = params, rng, eps
        self.W, self.H = params["width"], params["height"]
        self.terrain = params["terrain"]

    def cells_of(self, kind):
        return [c for c, t in enumerate(self.terrain) if t == kind]

    def step_towards(self, start, targets):
        goals = {}
        for tc in targets:
            tx, ty = tc % self.W, tc // self.W
            for a, (dx, dy) in DELTAS.items():
                nx, ny = tx - dx, ty - dy
                if 0 <= nx < self.W and 0 <= ny < self.H and self.terrain[ny * self.W + nx] == 0:
                    goals.setdefault(ny * self.W + nx, a)
        if not goals:
            return None
        if start in goals:
            return ("face", goals[start])
        seen, frontier = {start: None}, [start]
        while frontier:
            nxt = []
            for c in frontier:
                for a, (dx, dy) in DELTAS.items():
                    n = c + dx + dy * self.W
                    if self.terrain[n] != 0 or n in seen:
                        continue
                    seen[n] = (c, a)
                    if n in goals:
                        while seen[n][0] != start:
                            n = seen[n][0]
                        return ("move", seen[n][1])
                    nxt.append(n)
            frontier = nxt
        return None

    def act(self, mdp, state, who):
        if self.rng.random() < self.eps:
            return int(self.rng.integers(0, 6))
        pl = state.players[who]
        pots = self.cells_of(POT)
        held = pl.held_object
        soup_in = {c: state.objects[c] for c in pots if state.objects[c] != 0}
        if held == 0:
            busy = [c for c, s in soup_in.items() if s._cooking_tick >= 0]
            if busy and not any(q.held_object != 0 and q.held_object.name == 3 for q in state.players):
                targets = self.cells_of(DISH_SRC)
            else:
                srcs = self.cells_of(ONION_SRC) + self.cells_of(TOMATO_SRC)
                targets = [srcs[int(self.rng.integers(0, len(srcs)))]] if srcs else []
        elif held.name in (1, 2):
            targets = [c for c in pots if c not in soup_in or
                       (soup_in[c]._cooking_tick < 0 and soup_in[c].num_ingredients() < 3)] or self.cells_of(COUNTER)[:3]
        elif held.name == 3:
            ready = [c for c, s in soup_in.items() if mdp.is_ready(s)]
            if not ready and self.rng.random() < 0.15:
                targets = self.cells_of(COUNTER)[:4]       # park the dish: num_dishes_out bookkeeping
            else:
                targets = ready or pots
        else:
            targets = self.cells_of(SERVING)
        plan = self.step_towards(pl.position, targets)
        if plan is None:
            return int(self.rng.integers(0, 5))
        kind, a = plan
        if kind == "face":
            return 5 if pl.orientation == a else a
        return a


def make_stream(params, steps, seed, mode):
    rng = np.random.default_rng(seed)
    if isinstance(mode, str) and mode.startswith("cook:"):
        return np.zeros((steps, params["num_players"]), np.int8), Cook(params, rng, float(mode.split(":")[1]))
    return sample_actions(rng, steps, params["num_players"], mode), None


def rollout(params, actions, cook=None):
    """Lock-step reference rollout -> obs (T+1,P,C,F) uint8, reward (T,) int32, done (T,) int32."""
    mdp = DummyMDP(**params)
    state = mdp.get_standard_start_state()
    enc = lambda s: np.stack(mdp.lossless_state_encoding(s)).astype(np.uint8)
    obs = [enc(state)]
    rewards, dones = [], []
    for t, a in enumerate(actions):
        if cook is not None:
            a = actions[t] = np.array([cook.act(mdp, state, who) for who in range(params["num_players"])], np.int8)
        state, rew = mdp.get_state_transition(state, [int(x) for x in a])
        done = state.timestep >= params["horizon"]
        rewards.append(int(sum(rew)))
        dones.append(int(done))
        if done:
            state = mdp.get_standard_start_state()
        obs.append(enc(state))
    return np.stack(obs), np.array(rewards, np.int32), np.array(dones, np.int32)


def check_layout_transform():
    """Reference overcooked2 get_base_layout_params vs this repo's, on every old-style layout held here."""
    for name in ("gym", "gym.spaces", "oldercooked_ai_py", "oldercooked_ai_py.data", "oldercooked_ai_py.data.layouts",
                 "oldercooked_ai_py.utils", "oldercooked_ai_py.mdp", "oldercooked_ai_py.mdp.actions",
                 "oldercooked_ai_py.mdp.overcooked_mdp", "oldercooked_ai_py.mdp.overcooked_env", "build",
                 "build.madrona_simplecooked_example_python", "tensorboard", "torch.utils.tensorboard"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    sys.modules["gym"].Env = type("Env", (), {})
    for cls in ("Space", "Discrete", "MultiBinary", "Box", "MultiDiscrete"):
        setattr(sys.modules["gym.spaces"], cls, type(cls, (), {}))
    ref_dir = "/root/reference/oldercooked_ai/oldercooked_ai_py/data/layouts"
    read = lambda path: ast.literal_eval(open(path).read())
    sys.modules["oldercooked_ai_py.data.layouts"].read_layout_dict = lambda name: read(os.path.join(ref_dir, name + ".layout"))
    sys.modules["oldercooked_ai_py.utils"].load_dict_from_file = read
    sys.modules["oldercooked_ai_py.mdp.actions"].Action = type("Action", (), {"ALL_ACTIONS": list(range(6))})
    sys.modules["oldercooked_ai_py.mdp.overcooked_mdp"].OvercookedGridworld = object
    sys.modules["oldercooked_ai_py.mdp.overcooked_env"].OvercookedEnv = object
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    import envs.overcooked2_env as ref_env  # reference

    for name in layouts.SIMPLECOOKED_LAYOUTS:
        for cap in (None, 1):
            ref = ref_env.get_base_layout_params(name, 200, max_num_players=cap)
            mine = layouts.get_simplecooked_layout_params(name, 200, max_num_players=cap)
            assert ref == mine, (name, cap, ref, mine)
    print("simplecooked layout transform: identical to the reference on", len(layouts.SIMPLECOOKED_LAYOUTS), "layouts x 2 player caps")


def case(spec):
    layout, horizon, steps, seed, mode = spec[:5]
    cap = spec[5] if len(spec) > 5 else None
    return layout, horizon, steps, seed, mode, cap


def tomato_mask(params):
    """(C, F) bool: the one byte per tomato-source cell where the numpy twin and the C++ differ."""
    P, C = params["num_players"], params["height"] * params["width"]
    m = np.zeros((C, 5 * P + 10), bool)
    for c, t in enumerate(params["terrain"]):
        if t == TOMATO_SRC:
            m[c, 5 * P + 5] = True
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--soak", type=int, default=0, help="steps of extra oracle-vs-reference lock-step per case")
    args = ap.parse_args()
    check_layout_transform()

    if args.soak:
        from oracle.oracle import SimplecookedOracle
        for fixture, spec in CASES.items():
            layout, horizon, _, seed, mode, cap = case(spec)
            params = layouts.get_simplecooked_layout_params(layout, horizon, max_num_players=cap)
            acts, cook = make_stream(params, args.soak, 1000 + seed, mode)
            obs, rew, done = rollout(params, acts, cook)
            keep = ~tomato_mask(params)
            orc = SimplecookedOracle(params, 1)
            assert np.array_equal(orc.obs[0][:, keep], obs[0][:, keep]), fixture
            for t in range(args.soak):
                orc.step(acts[t].astype(np.int32)[:, None])
                assert np.array_equal(orc.obs[0][:, keep], obs[t + 1][:, keep]), (fixture, t)
                assert orc.reward[:, 0].tolist() == [rew[t]] * params["num_players"], (fixture, t, orc.reward[:, 0], rew[t])
                assert orc.done[0] == done[t], (fixture, t)
            print(f"soak {fixture}: {args.soak} steps identical, reward events {int((rew != 0).sum())}, total reward {int(rew.sum())}")
        return

    for fixture, spec in CASES.items():
        layout, horizon, steps, seed, mode, cap = case(spec)
        params = layouts.get_simplecooked_layout_params(layout, horizon, max_num_players=cap)
        acts, cook = make_stream(params, steps, seed, mode)
        obs, rew, done = rollout(params, acts, cook)
        out = os.path.join(HERE, f"simplecooked_{fixture}.npz")
        np.savez_compressed(out, params=json.dumps(params), actions=acts, obs=obs, reward=rew, done=done,
                            differs=tomato_mask(params))
        print(f"{fixture}: {steps} steps, obs {obs.shape}, reward events {int((rew != 0).sum())}, total {int(rew.sum())}, "
              f"resets {int(done.sum())}, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
