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


class VectorMultiAgentEnv(ABC):
    """N parallel worlds of an ``n_players`` game seen from one "ego" player;
    the other players are driven by registered partner agents."""

    def __init__(self, num_envs: int, device: torch.device, ego_ind: int = 0, n_players: int = 2,
                 resample_policy: str = "default", partners: Optional[List[List[VectorAgent]]] = None):
        self.num_envs = num_envs
        self.device = device
        self.ego_ind = ego_ind
        self.n_players = n_players
        if partners is not None:
            if len(partners) != n_players - 1:
                raise PlayerException("The number of partners needs to equal the number of non-ego players")
            for plist in partners:
                if not isinstance(plist, list) or not plist:
                    raise PlayerException("Sublist for each partner must be nonempty list")
        # NB: like the reference (`[[]] * (n - 1)`) the default lists are one shared object
        self.partners = partners or [[]] * (n_players - 1)
        self.partnerids = [0] * (n_players - 1)
        self._obs = tuple()
        self._actions = None
        self.set_resample_policy(resample_policy)

    def getDummyEnv(self, player_num: int):  # noqa: N802  (reference name)
        return self

    def _get_partner_num(self, player_num: int) -> int:
        if player_num == self.ego_ind:
            raise PlayerException("Ego agent is not set by the environment")
        return player_num - 1 if player_num > self.ego_ind else player_num

    def add_partner_agent(self, agent: VectorAgent, player_num: int = 1) -> None:
        self.partners[self._get_partner_num(player_num)].append(agent)

    def set_partnerid(self, agent_id: int, player_num: int = 1) -> None:
        partner_num = self._get_partner_num(player_num)
        assert 0 <= agent_id < len(self.partners[partner_num])
        self.partnerids[partner_num] = agent_id

    def resample_random(self) -> None:
        self.partnerids = [np.random.randint(len(plist)) for plist in self.partners]

    def resample_round_robin(self) -> None:
        self.partnerids = [(self.partnerids[0] + 1) % len(self.partners[0])]

    def set_resample_policy(self, resample_policy: str) -> None:
        if resample_policy == "default":
            resample_policy = "robin" if self.n_players == 2 else "random"
        if resample_policy == "robin" and self.n_players != 2:
            raise PlayerException("Cannot do round robin resampling for >2 players")
        if resample_policy == "robin":
            self.resample_partner = self.resample_round_robin
        elif resample_policy == "random":
            self.resample_partner = self.resample_random
        else:
            raise PlayerException(f"Invalid resampling policy: {resample_policy}")

    def _get_actions(self, obs, ego_act=None):
        actions = []
        for player, ob in zip(range(self.n_players), obs):
            if player == self.ego_ind:
                actions.append(ego_act)
            else:
                p = self._get_partner_num(player)
                actions.append(self.partners[p][self.partnerids[p]].get_action(ob))
        if self._actions is None:
            self._actions = torch.stack(actions)
        else:
            torch.stack(actions, out=self._actions)
        return self._actions

    def _update_players(self, rews, done):
        for i in range(self.n_players - 1):
            playernum = i + (0 if i < self.ego_ind else 1)
            self.partners[i][self.partnerids[i]].update(rews[playernum], done)

    def step(self, action: torch.Tensor):
        """One timestep from the ego player's point of view -> (obs, reward, done, info)."""
        acts = self._get_actions(self._obs, action)
        self._obs, rews, done, info = self.n_step(acts)
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
