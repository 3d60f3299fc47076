"""Vector multi-agent env API: behaviour of the reference's
``VectorMultiAgentEnv`` and ``MadronaEnv``
(/root/reference/pantheonrl_extension/vectorenv.py:26-255, 262-346) with the
same public names, so agents/trainers written against the reference run
unchanged.

``MadronaEnv`` differs in mechanism only: the reference re-indexes every export
through the agent/world id tensors with five ``index_put`` ops per step
(vectorenv.py:313-317); this engine exports (players, worlds, ...) tensors whose
memory order already is the declared order, so the "scattered" tensors *are*
the simulator's buffers and a step is: one action copy, one kernel pair, views.
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any, List, Optional

import numpy as np
import torch

from .vectoragent import VectorAgent
from .vectorobservation import VectorObservation


class PlayerException(Exception):
    """Raised when the players of an environment are set up inconsistently."""


@dataclass
class DummyEnv:
    """Observation/action spaces a partner agent needs to build its policy."""
    observation_space: Any
    action_space: Any


class PartnerRoster:
    """Who plays the non-ego seats.  Seat ``k`` (0-based among the non-ego players) has a list of candidate agents
    and a cursor into it; the env asks the roster for the seat's current agent, and a resampling policy moves the
    cursors between episodes.  This is the bookkeeping the reference keeps inline in ``VectorMultiAgentEnv``
    (vectorenv.py:39-135), pulled out so that the env class is only the step protocol."""

    POLICIES = ("robin", "random")

    def __init__(self, n_players: int, ego_ind: int, candidates: Optional[List[List[VectorAgent]]] = None):
        self.n_players, self.ego_ind = n_players, ego_ind
        seats = n_players - 1
        if candidates is not None:
            if len(candidates) != seats:
                raise PlayerException("The number of partners needs to equal the number of non-ego players")
            if any(not isinstance(c, list) or not c for c in candidates):
                raise PlayerException("Sublist for each partner must be nonempty list")
        # like the reference's `[[]] * (n - 1)`, the default seats share ONE list object (a partner added for
        # player 1 of a 3-player game is also a candidate for player 2)
        self.candidates = candidates or [[]] * seats
        self.cursor = [0] * seats

    def seat_of(self, player_num: int) -> int:
        if player_num == self.ego_ind:
            raise PlayerException("Ego agent is not set by the environment")
        return player_num - (1 if player_num > self.ego_ind else 0)

    def player_of(self, seat: int) -> int:
        return seat + (1 if seat >= self.ego_ind else 0)

    def current(self, seat: int) -> VectorAgent:
        return self.candidates[seat][self.cursor[seat]]

    def add(self, agent: VectorAgent, player_num: int) -> None:
        self.candidates[self.seat_of(player_num)].append(agent)

    def choose(self, agent_id: int, player_num: int) -> None:
        seat = self.seat_of(player_num)
        assert 0 <= agent_id < len(self.candidates[seat])
        self.cursor[seat] = agent_id

    def advance(self, policy: str) -> None:
        if policy == "random":
            self.cursor = [int(np.random.randint(len(c))) for c in self.candidates]
        else:  # round robin, two-player games only: one seat
            self.cursor = [(self.cursor[0] + 1) % len(self.candidates[0])]

    def resolve_policy(self, name: str) -> str:
        if name == "default":
            name = "robin" if self.n_players == 2 else "random"
        if name not in self.POLICIES:
            raise PlayerException(f"Invalid resampling policy: {name}")
        if name == "robin" and self.n_players != 2:
            raise PlayerException("Cannot do round robin resampling for >2 players")
        return name


class VectorMultiAgentEnv(ABC):
    """N parallel worlds of an ``n_players`` game seen from one "ego" player; the other players are driven by
    registered partner agents.  Public names and behaviour are the reference's (vectorenv.py:26-255): ``step`` /
    ``reset`` for the ego, ``n_step`` / ``n_reset`` for all players, ``add_partner_agent``, ``set_partnerid``,
    ``resample_*``, ``partners`` / ``partnerids``."""

    def __init__(self, num_envs: int, device: torch.device, ego_ind: int = 0, n_players: int = 2,
                 resample_policy: str = "default", partners: Optional[List[List[VectorAgent]]] = None):
        self.num_envs = num_envs
        self.device = device
        self.ego_ind = ego_ind
        self.n_players = n_players
        self._roster = PartnerRoster(n_players, ego_ind, partners)
        self._obs = tuple()
        self._actions = None
        self.set_resample_policy(resample_policy)

    # the reference exposes these two as plain attributes; callers read and occasionally assign them
    @property
    def partners(self):
        return self._roster.candidates

    @partners.setter
    def partners(self, value):
        self._roster.candidates = value

    @property
    def partnerids(self):
        return self._roster.cursor

    @partnerids.setter
    def partnerids(self, value):
        self._roster.cursor = list(value)

    def getDummyEnv(self, player_num: int):  # noqa: N802  (reference name)
        return self

    def _get_partner_num(self, player_num: int) -> int:
        self._roster.ego_ind = self.ego_ind  # subclasses set ego_ind after construction
        return self._roster.seat_of(player_num)

    def add_partner_agent(self, agent: VectorAgent, player_num: int = 1) -> None:
        self._roster.ego_ind = self.ego_ind
        self._roster.add(agent, player_num)

    def set_partnerid(self, agent_id: int, player_num: int = 1) -> None:
        self._roster.ego_ind = self.ego_ind
        self._roster.choose(agent_id, player_num)

    def resample_random(self) -> None:
        self._roster.advance("random")

    def resample_round_robin(self) -> None:
        self._roster.advance("robin")

    def set_resample_policy(self, resample_policy: str) -> None:
        policy = self._roster.resolve_policy(resample_policy)
        self.resample_partner = self.resample_round_robin if policy == "robin" else self.resample_random

    def _get_actions(self, obs, ego_act=None):
        """Joint action (n_players, num_envs, 1): the ego's plus what each seat's current agent answers."""
        self._roster.ego_ind = self.ego_ind
        joint = [ego_act if player == self.ego_ind else self._roster.current(self._roster.seat_of(player)).get_action(ob)
                 for player, ob in zip(range(self.n_players), obs)]
        if self._actions is None:
            self._actions = torch.stack(joint)
        else:
            torch.stack(joint, out=self._actions)
        return self._actions

    def _update_players(self, rews, done):
        self._roster.ego_ind = self.ego_ind
        for seat in range(self.n_players - 1):
            self._roster.current(seat).update(rews[self._roster.player_of(seat)], done)

    def step(self, action: torch.Tensor, out: Optional[torch.Tensor] = None):
        """One timestep from the ego player's point of view -> (obs, reward, done, info).  ``out`` (extension): handed
        to ``n_step`` of environments that can write a step's observations into a caller's buffer slot."""
        joint = self._get_actions(self._obs, action)
        self._obs, rews, done, info = self.n_step(joint) if out is None else self.n_step(joint, out=out)
        self._update_players(rews, done)
        return self._obs[self.ego_ind], rews[self.ego_ind], done, info

    def reset(self):
        """Resample partners and return the ego player's current observation.
        (Worlds restart on their own inside ``step``; see ``n_reset``.)"""
        self.resample_partner()
        self._obs = self.n_reset()
        return self._obs[self.ego_ind]

    @abstractmethod
    def n_step(self, actions: torch.Tensor):
        """actions (n_players, num_envs, 1) -> (list of VectorObservation, rewards
        (n_players, num_envs), dones (num_envs,), infos)."""

    @abstractmethod
    def n_reset(self):
        """-> list of VectorObservation, one per player."""

    def close(self, **kwargs):
        pass


class MadronaEnv(VectorMultiAgentEnv):
    """Generic wrapper over a simulator exporting (players, worlds, ...) tensors;
    used by Hanabi (/root/reference/envs/hanabi_env.py:72-104)."""

    def __init__(self, num_envs, gpu_id, sim, debug_compile=True, obs_size=None, state_size=None,
                 discrete_action_size=None, env_device=None):
        self.sim = sim
        self.static_dones = sim.done_tensor().to_torch()
        self.static_active_agents = sim.active_agent_tensor().to_torch()
        self.static_actions = sim.action_tensor().to_torch()
        self.static_observations = sim.observation_tensor().to_torch()
        self.static_agent_states = sim.agent_state_tensor().to_torch()
        self.static_action_masks = sim.action_mask_tensor().to_torch()
        self.static_rewards = sim.reward_tensor().to_torch()
        self.static_worldID = sim.world_id_tensor().to_torch().to(torch.long)
        self.static_agentID = sim.agent_id_tensor().to_torch().to(torch.long)

        self.obs_size = self.static_observations.shape[2] if obs_size is None else obs_size
        self.state_size = self.static_agent_states.shape[2] if state_size is None else state_size
        self.discrete_action_size = (self.static_action_masks.shape[2] if discrete_action_size is None
                                     else discrete_action_size)
        # memory order == declared order here, so the reference's "scattered" copies alias the exports
        self.static_scattered_active_agents = self.static_active_agents
        self.static_scattered_observations = self.static_observations
        self.static_scattered_agent_states = self.static_agent_states
        self.static_scattered_action_masks = self.static_action_masks
        self.static_scattered_rewards = self.static_rewards

        if env_device is None:
            env_device = self.static_observations.device
        super().__init__(num_envs, device=env_device, n_players=self.static_observations.shape[0])
        self.infos = [{}] * self.num_envs

    def to_torch(self, a):
        return a.to(self.device)

    def _observations(self):
        return [VectorObservation(self.to_torch(self.static_active_agents[i].to(torch.bool)),
                                  self.to_torch(self.static_observations[i, :, :self.obs_size]),
                                  self.to_torch(self.static_agent_states[i, :, :self.state_size]),
                                  self.to_torch(self.static_action_masks[i, :, :self.discrete_action_size].to(torch.bool)))
                for i in range(self.n_players)]

    def n_step(self, actions):
        self.static_actions.copy_(actions.to(self.static_actions.device), non_blocking=True)
        self.sim.step()
        return self._observations(), self.to_torch(self.static_rewards), self.to_torch(self.static_dones), self.infos

    def n_reset(self):
        return self._observations()


class SyncVectorEnv(VectorMultiAgentEnv):
    """The baseline the reference measures its batched simulators against: N ordinary single-world multi-agent
    environments stepped one after the other on the host, presented through the vector API
    (/root/reference/pantheonrl_extension/vectorenv.py:348-457).  ``env_fns``: callables that build the environments;
    each environment follows the reference's ``MultiAgentEnv`` protocol --

        ``n_reset() -> (agents, observations)``
        ``n_step(actions) -> (agents, observations, rewards, done, info)``

    where ``agents`` are the players to act next, ``observations[j] = (obs, state, action_mask)`` numpy arrays for
    ``agents[j]`` and ``actions`` holds one action per player that was asked to act.  A finished environment is reset at
    once and its first observation of the next episode returned, with the last step's rewards and ``done`` (the batched
    simulators behave the same way).  Players that are not to act keep their previous rows and are inactive.

    Nothing of the HIP engine is involved: this is host-side plumbing for environments that exist only in Python.
    (Mechanism: the reference writes every scalar straight into device tensors, several tiny copies per player and
    world and step; here a step is assembled in host arrays and crosses to the device once per tensor.)"""

    def __init__(self, env_fns, device=None):
        if device is None:
            device = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        self.envs = [fn() for fn in env_fns]
        if not self.envs:
            raise ValueError("SyncVectorEnv needs at least one environment")
        first = self.envs[0]
        self.observation_space = first.observation_space
        self.action_space = first.action_space
        self.share_observation_space = first.share_observation_space
        super().__init__(len(self.envs), device=device, n_players=first.n_players)
        self.agents_tuples = []
        self._host = None  # host-side staging arrays, allocated from the first observation's shapes

    def _allocate(self, sample):
        obs, state, mask = (np.asarray(part) for part in sample)
        P, N = self.n_players, self.num_envs
        self._host = {
            "active": np.zeros((P, N), dtype=bool),
            "obs": np.zeros((P, N) + obs.shape, dtype=np.float32),
            "state": np.zeros((P, N) + state.shape, dtype=np.float32),
            "mask": np.ones((P, N) + mask.shape, dtype=bool),
        }

    def _place(self, world, agents, observations):
        host = self._host
        host["active"][:, world] = False
        for agent, (obs, state, mask) in zip(agents, observations):
            host["active"][agent, world] = True
            host["obs"][agent, world] = obs
            host["state"][agent, world] = state
            host["mask"][agent, world] = mask

    def _publish(self):
        host = self._host
        self.static_active_agents = torch.from_numpy(host["active"]).to(self.device)
        self.static_observations = torch.from_numpy(host["obs"]).to(self.device)
        self.static_agent_states = torch.from_numpy(host["state"]).to(self.device)
        self.static_action_masks = torch.from_numpy(host["mask"]).to(self.device)
        return [VectorObservation(self.static_active_agents[p], self.static_observations[p], self.static_agent_states[p],
                                  self.static_action_masks[p]) for p in range(self.n_players)]

    def n_reset(self):
        self.agents_tuples = []
        for world, env in enumerate(self.envs):
            agents, observations = env.n_reset()
            if self._host is None:
                self._allocate(observations[0])
            self.agents_tuples.append(tuple(agents))
            self._place(world, agents, observations)
        return self._publish()

    def n_step(self, actions):
        """``actions``: (n_players, num_envs, ...) -- only the entries of the players that were asked to act are used."""
        acts = actions.detach().cpu().numpy() if torch.is_tensor(actions) else np.asarray(actions)
        rewards = np.zeros((self.n_players, self.num_envs), dtype=np.float32)
        dones = np.zeros(self.num_envs, dtype=bool)
        infos = []
        for world, env in enumerate(self.envs):
            asked = tuple(acts[agent, world] for agent in self.agents_tuples[world])
            agents, observations, rews, done, info = env.n_step(asked)
            if done:
                agents, observations = env.n_reset()
            self.agents_tuples[world] = tuple(agents)
            rewards[:, world] = np.asarray(rews, dtype=np.float32)[:self.n_players]
            dones[world] = bool(done)
            infos.append(info)
            self._place(world, agents, observations)
        self.static_rewards = torch.from_numpy(rewards).to(self.device)
        self.static_dones = torch.from_numpy(dones).to(self.device)
        return self._publish(), self.static_rewards, self.static_dones, infos

    def close(self, **kwargs):
        for env in self.envs:
            if hasattr(env, "close"):
                env.close()
