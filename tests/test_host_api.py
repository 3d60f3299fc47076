"""CPU: host-side logic of the vector-env API (ego/partner plumbing, resampling,
spaces) with a fake env -- mirrors the behaviour of the reference's
VectorMultiAgentEnv (pantheonrl_extension/vectorenv.py:26-255)."""
import numpy as np
import pytest
import torch

from madrona_rl_envs_playground_amd import spaces
from madrona_rl_envs_playground_amd.pantheonrl_extension import (PlayerException, RandomVectorAgent, VectorAgent,
                                                                 VectorMultiAgentEnv, VectorObservation)
from madrona_rl_envs_playground_amd.simulators import ExecMode, madrona


class CountingEnv(VectorMultiAgentEnv):
    """obs of player p = step counter + p; reward of player p = sum of actions + p."""

    def __init__(self, n_envs, n_players=2, ego_ind=0):
        super().__init__(n_envs, torch.device("cpu"), ego_ind=ego_ind, n_players=n_players)
        self.t = 0
        self.seen_actions = None

    def _obs_list(self):
        return [VectorObservation(torch.ones(self.num_envs, dtype=torch.bool),
                                  torch.full((self.num_envs, 3), self.t + p)) for p in range(self.n_players)]

    def n_step(self, actions):
        self.seen_actions = actions.clone()
        self.t += 1
        rew = torch.stack([actions.sum(0).float().squeeze(-1) + p for p in range(self.n_players)])
        return self._obs_list(), rew, torch.zeros(self.num_envs, dtype=torch.int32), [{}] * self.num_envs

    def n_reset(self):
        return self._obs_list()


class Recorder(VectorAgent):
    def __init__(self, value, n):
        self.value, self.n, self.updates, self.seen = value, n, [], []

    def get_action(self, obs, record=True):
        self.seen.append(obs.obs.clone())
        return torch.full((self.n, 1), self.value)

    def update(self, rewards, dones):
        self.updates.append((rewards.clone(), dones.clone()))


def test_vector_observation_defaults():
    o = VectorObservation(torch.ones(4, dtype=torch.bool), torch.zeros(4, 7))
    assert o.state is o.obs and o.action_mask is None
    s = torch.ones(4, 9)
    assert VectorObservation(o.active, o.obs, s).state is s


def test_ego_step_drives_partners():
    env = CountingEnv(5)
    partner = Recorder(3, 5)
    env.add_partner_agent(partner, player_num=1)
    first = env.reset()
    assert torch.equal(first.obs, torch.zeros(5, 3))
    ob, rew, done, info = env.step(torch.full((5, 1), 2))
    assert env.seen_actions.shape == (2, 5, 1)
    assert (env.seen_actions[0] == 2).all() and (env.seen_actions[1] == 3).all()
    assert torch.equal(partner.seen[0], torch.ones(5, 3))          # partner saw its own (player-1) observation
    assert torch.equal(rew, torch.full((5,), 5.0))                  # ego reward = sum of actions + 0
    assert torch.equal(partner.updates[0][0], torch.full((5,), 6.0))
    assert torch.equal(ob.obs, torch.ones(5, 3)) and len(info) == 5


def test_ego_index_and_three_players():
    env = CountingEnv(2, n_players=3, ego_ind=1)
    a, b = Recorder(1, 2), Recorder(4, 2)
    env.add_partner_agent(a, player_num=0)
    with pytest.raises(PlayerException):
        env.add_partner_agent(b, player_num=1)      # the ego slot
    # like the reference, the default partner lists are one shared list object
    assert env.partners[0] is env.partners[1]
    env2 = CountingEnv(2, n_players=3, ego_ind=1)
    env2.partners = [[a], [b]]
    env2.reset()
    env2.step(torch.full((2, 1), 9))
    assert env2.seen_actions[:, 0, 0].tolist() == [1, 9, 4]


def test_resample_policies():
    env = CountingEnv(1)
    for v in range(3):
        env.add_partner_agent(Recorder(v, 1))
    ids = []
    for _ in range(4):
        env.reset()
        ids.append(env.partnerids[0])
    assert ids == [1, 2, 0, 1]                       # round robin for two players
    with pytest.raises(PlayerException):
        CountingEnv(1, n_players=3).set_resample_policy("robin")
    with pytest.raises(PlayerException):
        env.set_resample_policy("nonsense")
    env.set_resample_policy("random")
    np.random.seed(0)
    env.reset()
    assert 0 <= env.partnerids[0] < 3
    env.set_partnerid(2)
    assert env.partnerids == [2]
    with pytest.raises(PlayerException):
        VectorMultiAgentEnv.__init__(env, 1, torch.device("cpu"), partners=[[], []])


def test_random_agent_and_spaces():
    agent = RandomVectorAgent(lambda: torch.zeros(3, 1))
    assert agent.get_action(None).shape == (3, 1) and agent.update(None, None) is None
    d = spaces.Discrete(6)
    assert d.n == 6 and 0 <= d.sample() < 6
    mb = spaces.MultiBinary(np.array([5, 4, 26]))
    assert tuple(mb.shape) == (5, 4, 26)
    assert spaces.MultiBinary(658).shape == (658,)
    box = spaces.Box(-np.ones(4, np.float32), np.ones(4, np.float32), dtype=np.float32)
    assert box.shape == (4,) and box.sample().shape == (4,)


def test_exec_mode_shim():
    assert madrona.ExecMode.CPU is ExecMode.CPU and madrona.ExecMode.CUDA is ExecMode.CUDA
    assert ExecMode.HIP is ExecMode.CUDA


def test_random_action_stream_is_uniform_and_reproducible():
    """The device random policy's hash (include/mrl_envs.h, mrl_rollout_random) as restated on the host."""
    from madrona_rl_envs_playground_amd.simulators import random_action
    world, player = np.meshgrid(np.arange(50000), np.arange(2))
    a = random_action(7, 3, world, player)
    assert a.dtype == np.int32 and a.shape == (2, 50000) and a.min() == 0 and a.max() == 5
    freq = np.bincount(a.ravel(), minlength=6) / a.size
    assert np.abs(freq - 1 / 6).max() < 0.01
    assert np.array_equal(a, random_action(7, 3, world, player))
    assert (a != random_action(7, 4, world, player)).mean() > 0.7
    assert (a != random_action(8, 3, world, player)).mean() > 0.7
    assert (a[0] != a[1]).mean() > 0.7
    assert (a != random_action(7 + (1 << 32), 3, world, player)).mean() > 0.7
    # known answers, computed by hand-evaluating the formula in the header with Python integers
    def scalar(seed, k, w, q):
        m = 0xFFFFFFFF
        h = (seed & m) ^ (k * 0x9E3779B9 & m) ^ (w * 0x85EBCA6B & m) ^ ((q + 1) * 0xC2B2AE35 & m) ^ ((seed >> 32) * 0x27D4EB2F & m)
        h ^= h >> 16; h = h * 0x7FEB352D & m; h ^= h >> 15; h = h * 0x846CA68B & m; h ^= h >> 16
        return (h * 6) >> 32
    for seed, k, w, q in [(0, 0, 0, 0), (7, 3, 49999, 1), (2 ** 64 - 1, 2 ** 32 - 1, 12345, 63)]:
        assert int(random_action(seed, k, np.array([w]), np.array([q]))[0]) == scalar(seed, k, w, q)


def test_random_policy_host_restatements():
    """Cartpole and Hanabi draws as documented in include/mrl_envs.h (mrl_rollout_random)."""
    from madrona_rl_envs_playground_amd.simulators import random_cartpole_action, random_hanabi_action, random_hash
    w = np.arange(20000)
    a = random_cartpole_action(3, 11, w)
    assert set(np.unique(a)) == {0, 1} and abs(a.mean() - 0.5) < 0.02
    assert np.array_equal(a, (random_hash(3, 11, w, np.zeros_like(w)) >> 31).astype(np.int32))
    rng = np.random.default_rng(0)
    legal = (rng.random((20000, 20)) < 0.4).astype(np.int32)
    legal[:5] = 0                      # no legal move: action 0 by convention
    legal[5] = 0
    legal[5, 19] = 1
    mover = rng.integers(0, 2, 20000)
    act = random_hanabi_action(5, 2, w, mover, legal)
    assert (act[:5] == 0).all() and act[5] == 19
    ok = legal.sum(-1) > 0
    assert (legal[w[ok], act[ok]] == 1).all()
    # uniform over the legal moves: position among the legal ones is uniform
    rank = (np.cumsum(legal, -1) - 1)[w[ok], act[ok]] / legal.sum(-1)[ok]
    assert abs(rank.mean() - (0.5 - (0.5 / legal.sum(-1)[ok]).mean())) < 0.02


class TurnTaker:
    """A single-world two-player environment in the reference's MultiAgentEnv protocol
    (pantheonrl_extension/multiagentenv.py:236-276): players alternate, an episode lasts `length` moves, the observation is
    (move counter, mover), the reward of both players is the action taken."""
    n_players = 2
    observation_space = spaces.MultiBinary(2)
    share_observation_space = spaces.MultiBinary(3)
    action_space = spaces.Discrete(4)

    def __init__(self, length, offset):
        self.length, self.offset, self.t, self.episodes, self.log = length, offset, 0, 0, []

    def _view(self):
        mover = (self.t + self.offset) % 2
        obs = np.array([self.t, mover], np.float32)
        return (mover,), ((obs, np.array([self.t, mover, self.episodes], np.float32), np.array([1, 1, mover, 1], bool)),)

    def n_reset(self):
        self.t = 0
        self.episodes += 1
        return self._view()

    def n_step(self, actions):
        assert len(actions) == 1
        self.log.append(int(np.asarray(actions[0]).reshape(-1)[0]))
        self.t += 1
        agents, obs = self._view()
        rew = float(self.log[-1])
        return agents, obs, (rew, rew), self.t >= self.length, {"t": self.t}


def test_sync_vector_env_steps_python_envs_one_by_one():
    """SyncVectorEnv (reference vectorenv.py:348-457): per-world resets inside n_step, inactive players keep their rows,
    only the acting player's action reaches an environment."""
    from madrona_rl_envs_playground_amd.pantheonrl_extension import SyncVectorEnv
    lengths = [3, 5, 2]
    env = SyncVectorEnv([lambda k=k, n=n: TurnTaker(n, k) for k, n in enumerate(lengths)], device=torch.device("cpu"))
    assert env.num_envs == 3 and env.n_players == 2 and env.action_space.n == 4
    obs = env.n_reset()
    assert len(obs) == 2 and obs[0].obs.shape == (3, 2) and obs[0].state.shape == (3, 3) and obs[0].action_mask.shape == (3, 4)
    assert obs[0].active.tolist() == [True, False, True] and obs[1].active.tolist() == [False, True, False]
    rng = np.random.default_rng(0)
    t = [0, 0, 0]
    for step in range(12):
        acts = torch.from_numpy(rng.integers(0, 4, size=(2, 3, 1)))
        movers = [int(obs[1].active[w]) for w in range(3)]
        obs, rew, done, infos = env.n_step(acts)
        for w in range(3):
            took = int(acts[movers[w], w, 0])
            assert env.envs[w].log[-1] == took                                   # the mover's action, nobody else's
            assert rew[:, w].tolist() == [took, took]
            t[w] += 1
            ended = t[w] >= lengths[w]
            assert bool(done[w]) == ended and infos[w]["t"] == t[w]
            if ended:
                t[w] = 0                                                         # the next episode's first observation
            mover = (t[w] + w) % 2
            assert obs[mover].active[w] and not obs[1 - mover].active[w]
            assert obs[mover].obs[w].tolist() == [t[w], mover]
            assert obs[mover].action_mask[w].tolist() == [True, True, bool(mover), True]
    assert [e.episodes for e in env.envs] == [1 + 12 // n for n in lengths]
    # the ego-perspective step on top of it
    env.add_partner_agent(RandomVectorAgent(lambda: torch.zeros((3, 1), dtype=torch.int64)))
    ego = env.reset()
    assert ego.obs.shape == (3, 2)
    ob, r, d, info = env.step(torch.ones((3, 1), dtype=torch.int64))
    assert ob.obs.shape == (3, 2) and r.shape == (3,) and d.shape == (3,)
    env.close()
