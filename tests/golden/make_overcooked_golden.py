"""Generates tests/golden/overcooked_*.npz from the REFERENCE's own numpy
implementation of Overcooked (envs/overcooked_reimplement.py, class DummyMDP),
imported from /root/reference in the build container.  Run only there:

    python tests/golden/make_overcooked_golden.py            # write fixtures
    python tests/golden/make_overcooked_golden.py --soak 20000   # extra lock-step check, nothing written

The fixtures hold data only: the simulator config, an action stream and the
observations / rewards / dones the reference produced for it.  Episode handling
follows the reference's validation loop (envs/overcooked_env.py:408-424,478-527):
done = timestep >= horizon, and on done the next observation is that of a fresh
start state.

It also cross-checks this repo's layout transform against the reference's
get_base_layout_params (imported with stub modules for the uninstalled
gym / overcooked_ai_py / build.* packages; ordinary ModuleNotFoundError, no
permission was denied).
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

from envs.overcooked_reimplement import DummyMDP  # noqa: E402  (reference)

# fixture name -> (layout, horizon, max players, steps, seed, P(interact) or None for uniform)
CASES = {
    "cramped_room": ("cramped_room", 400, None, 900, 0, None),
    "cramped_room_busy": ("cramped_room", 60, None, 400, 1, 0.45),
    "asymmetric_advantages": ("asymmetric_advantages", 400, None, 500, 2, None),
    "coordination_ring": ("coordination_ring", 400, None, 500, 3, 0.35),
    "forced_coordination": ("forced_coordination", 400, None, 500, 4, 0.35),
    "counter_circuit": ("counter_circuit", 400, None, 500, 5, None),
    "multiplayer_schelling": ("multiplayer_schelling", 100, None, 400, 6, 0.35),
    "tomato_mix": ("asymmetric_advantages_tomato", 120, None, 600, 7, 0.45),
    "many_player_8": ("many_player_layout", 50, 8, 120, 8, 0.35),
    # goal-directed streams (P(interact) slot = "cook:<eps>")
    "cramped_room_cook": ("cramped_room", 400, None, 900, 9, "cook:0.1"),
    "coordination_ring_cook": ("coordination_ring", 250, None, 600, 10, "cook:0.15"),
    "tomato_mix_cook": ("asymmetric_advantages_tomato", 200, None, 600, 11, "cook:0.1"),
    "multiplayer_schelling_cook": ("multiplayer_schelling", 150, None, 400, 12, "cook:0.2"),
}


def make_stream(params, steps, seed, mode):
    rng = np.random.default_rng(seed)
    if isinstance(mode, str) and mode.startswith("cook:"):
        acts = np.zeros((steps, params["num_players"]), np.int8)
        return acts, Cook(params, rng, float(mode.split(":")[1]))
    return sample_actions(rng, steps, params["num_players"], mode), None


def sample_actions(rng, steps, players, p_interact):
    if p_interact is None:
        return rng.integers(0, 6, size=(steps, players)).astype(np.int8)
    a = rng.integers(0, 5, size=(steps, players))
    a[rng.random((steps, players)) < p_interact] = 5
    return a.astype(np.int8)


DELTAS = {0: (0, -1), 1: (0, 1), 2: (1, 0), 3: (-1, 0)}  # action -> (dx, dy)


class Cook:
    """A small goal-directed policy (fetch ingredient -> pot -> start -> dish ->
    plate -> serve) with epsilon-random noise, so the fixtures contain complete
    soup cycles, which a uniform random policy almost never produces."""

    def __init__(self, params, rng, eps):
        self.p, self.rng, self.eps = params, rng, eps
        self.W, self.H = params["width"], params["height"]
        self.terrain = params["terrain"]

    def cells_of(self, kind):
        return [c for c, t in enumerate(self.terrain) if t == kind]

    def step_towards(self, start, targets):
        """First action of a shortest walk to an AIR cell next to one of ``targets`` (then face it)."""
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
        pots = self.cells_of(1)
        held = pl.held_object
        soup_in = {c: state.objects[c] for c in pots if state.objects[c] != 0}
        if held == 0:
            startable = [c for c, s in soup_in.items() if s._cooking_tick < 0 and s.num_ingredients() == 3]
            busy = [c for c, s in soup_in.items() if s._cooking_tick >= 0]
            if startable:
                targets = startable
            elif busy and not any(q.held_object != 0 and q.held_object.name == 3 for q in state.players):
                targets = self.cells_of(5)
            else:
                srcs = self.cells_of(3) + self.cells_of(4)
                targets = [srcs[int(self.rng.integers(0, len(srcs)))]] if srcs else []
        elif held.name in (1, 2):
            targets = [c for c in pots if c not in soup_in or
                       (soup_in[c]._cooking_tick < 0 and soup_in[c].num_ingredients() < 3)] or self.cells_of(2)[:3]
        elif held.name == 3:
            targets = [c for c, s in soup_in.items() if mdp.is_ready(s)] or pots
        else:
            targets = self.cells_of(6)
        plan = self.step_towards(pl.position, targets)
        if plan is None:
            return int(self.rng.integers(0, 5))
        kind, a = plan
        if kind == "face":
            return 5 if pl.orientation == a else a
        return a


def rollout(params, actions, cook=None):
    """Lock-step reference rollout -> obs (T+1,P,C,F) uint8, reward (T,) int32, done (T,) int32.
    With ``cook`` the action stream is produced on the fly (and written back into ``actions``)."""
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
    """Reference get_base_layout_params vs this repo's, on every layout."""
    for name in ("gym", "gym.spaces", "overcooked_ai_py", "overcooked_ai_py.utils", "overcooked_ai_py.mdp",
                 "overcooked_ai_py.mdp.actions", "overcooked_ai_py.mdp.overcooked_mdp",
                 "overcooked_ai_py.mdp.overcooked_env", "build", "build.madrona_overcooked_example_python",
                 "tensorboard", "torch.utils.tensorboard"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    sys.modules["gym"].Env = type("Env", (), {})
    for cls in ("Space", "Discrete", "MultiBinary", "Box", "MultiDiscrete"):
        setattr(sys.modules["gym.spaces"], cls, type(cls, (), {}))
    u = sys.modules["overcooked_ai_py.utils"]
    u.read_layout_dict = lambda name: {k: v for k, v in layouts.LAYOUTS[name].items()}
    u.load_dict_from_file = lambda path: ast.literal_eval(open(path).read())
    sys.modules["overcooked_ai_py.mdp.actions"].Action = type("Action", (), {"NUM_ACTIONS": 6})
    sys.modules["overcooked_ai_py.mdp.overcooked_mdp"].OvercookedGridworld = object
    sys.modules["overcooked_ai_py.mdp.overcooked_env"].OvercookedEnv = object
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    import envs.overcooked_env as ref_env  # reference

    for name in layouts.LAYOUTS:
        for cap in (None, 2, 3):
            ref = ref_env.get_base_layout_params(name, 400, max_num_players=cap)
            mine = layouts.get_base_layout_params(name, 400, max_num_players=cap)
            assert ref == mine, (name, cap, ref, mine)
    print("layout transform: identical to the reference on", len(layouts.LAYOUTS), "layouts x 3 player caps")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--soak", type=int, default=0, help="steps of extra oracle-vs-reference lock-step per layout")
    args = ap.parse_args()
    check_layout_transform()

    if args.soak:
        from oracle.oracle import OvercookedOracle
        for fixture, (layout, horizon, cap, _, seed, p_int) in CASES.items():
            params = layouts.get_base_layout_params(layout, horizon, max_num_players=cap)
            steps = args.soak if params["num_players"] <= 4 else max(200, args.soak // 20)
            acts, cook = make_stream(params, steps, 1000 + seed, p_int)
            obs, rew, done = rollout(params, acts, cook)
            orc = OvercookedOracle(params, 1)
            assert np.array_equal(orc.obs[0], obs[0]), fixture
            for t in range(steps):
                orc.step(acts[t].astype(np.int32)[:, None])
                assert np.array_equal(orc.obs[0], obs[t + 1]), (fixture, t)
                assert orc.reward[:, 0].tolist() == [rew[t]] * params["num_players"], (fixture, t)
                assert orc.done[0] == done[t], (fixture, t)
            print(f"soak {fixture}: {steps} steps identical, reward events {int((rew != 0).sum())}, "
                  f"total reward {int(rew.sum())}")
        return

    for fixture, (layout, horizon, cap, steps, seed, p_int) in CASES.items():
        params = layouts.get_base_layout_params(layout, horizon, max_num_players=cap)
        acts, cook = make_stream(params, steps, seed, p_int)
        obs, rew, done = rollout(params, acts, cook)
        out = os.path.join(HERE, f"overcooked_{fixture}.npz")
        np.savez_compressed(out, params=json.dumps(params), actions=acts, obs=obs, reward=rew, done=done)
        print(f"{fixture}: {steps} steps, obs {obs.shape}, reward events {int((rew != 0).sum())}, "
              f"resets {int(done.sum())}, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
