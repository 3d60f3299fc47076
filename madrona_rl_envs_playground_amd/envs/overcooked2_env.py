"""``OvercookedMadrona`` of the "Simplecooked" world -- drop-in for /root/reference/envs/overcooked2_env.py:19-115
(the class the reference's trainer binds, train/env_utils.py:3).

Same constructor (horizon defaults to 200 here), attributes (``static_actions``, ``static_observations``,
``static_action_mask``, ``sim`` ...) and return shapes: ``get_obs`` gives, per player, an ``(N, W, H, 5P+10)``
int8 view plus the all-ones ``(N, 6)`` action mask the reference attaches (overcooked2_env.py:55,96-99).  As in
``overcooked_env.py`` here, the HIP kernel writes a world-major ``(N, P, H, W, F)`` block, so no scatter runs.
"""
import numpy as np
import torch

from ..layouts import get_simplecooked_layout_params as get_base_layout_params  # noqa: F401  (reference name)
from ..pantheonrl_extension.vectorenv import VectorMultiAgentEnv
from ..pantheonrl_extension.vectorobservation import VectorObservation
from ..simulators import ExecMode, SimplecookedSimulator
from ..spaces import Discrete, MultiBinary

NUM_ACTIONS = 6  # oldercooked_ai_py Action.ALL_ACTIONS


class OvercookedMadrona(VectorMultiAgentEnv):

    def __init__(self, layout_name, num_envs, gpu_id, debug_compile=True, use_cpu=False, use_env_cpu=False,
                 ego_agent_idx=0, horizon=200, num_players=None):
        self.layout_name = layout_name
        self.base_layout_params = get_base_layout_params(layout_name, horizon, max_num_players=num_players)
        self.width = self.base_layout_params["width"]
        self.height = self.base_layout_params["height"]
        self.num_players = self.base_layout_params["num_players"]
        self.size = self.width * self.height
        self.horizon = horizon

        self.sim = SimplecookedSimulator(exec_mode=ExecMode.CPU if use_cpu else ExecMode.CUDA, gpu_id=gpu_id,
                                         num_worlds=num_envs, debug_compile=debug_compile, **self.base_layout_params)

        full_obs_size = self.width * self.height * (5 * self.num_players + 10)
        self.obs_size = full_obs_size
        self.state_size = full_obs_size

        self.static_dones = self.sim.done_tensor().to_torch()
        self.static_active_agents = self.sim.active_agent_tensor().to_torch().to(torch.bool)
        self.static_actions = self.sim.action_tensor().to_torch()
        self.static_observations = self.sim.observation_tensor().to_torch()
        self.static_rewards = self.sim.reward_tensor().to_torch()
        self.static_world_major_observations = self.sim.observation_world_major_tensor().to_torch()
        self.static_scattered_active_agents = self.static_active_agents
        self.static_scattered_rewards = self.static_rewards
        self.static_scattered_observations = self.static_observations
        self.static_action_mask = torch.ones((num_envs, NUM_ACTIONS), dtype=torch.bool, device=self.static_dones.device)

        env_device = torch.device("cpu") if use_env_cpu else self.static_dones.device
        super().__init__(num_envs, device=env_device, n_players=self.num_players)

        self.infos = [{}] * self.num_envs
        self.ego_ind = ego_agent_idx
        self.observation_space = self._setup_observation_space()
        self.share_observation_space = self.observation_space
        self.action_space = Discrete(NUM_ACTIONS)
        self._player_views = [self.static_world_major_observations[:, i].transpose(1, 2)
                              for i in range(self.num_players)]
        # On the simulator's own device ``.to(device)`` hands back the very same tensors: what get_obs / n_step return is
        # then put together once (the per-call host work is what bounds small batches, tools/host_overhead.py)
        self._resident = env_device == self.static_dones.device
        self._obs_parts = [(self.static_active_agents[i], self._player_views[i]) for i in range(self.num_players)]
        self._n_actions = self.static_actions.numel()
        self._redirected_to = 0  # data pointer the simulator's observation output is redirected to (n_step(out=...)), 0 = its own tensor
        self.n_reset()

    def _setup_observation_space(self):
        return MultiBinary(np.array([self.width, self.height, 5 * self.num_players + 10]))

    def to_torch(self, a):
        return a.to(self.device)

    def get_obs(self):
        if self._resident:
            return [VectorObservation(active, view, action_mask=self.static_action_mask) for active, view in self._obs_parts]
        return [VectorObservation(self.to_torch(active), self.to_torch(view), action_mask=self.static_action_mask) for active, view in self._obs_parts]

    def n_step(self, actions, out=None):
        """``out`` (extension, no reference counterpart): an int8 CUDA tensor of shape (N, P, H, W, F) -- one slot of a
        rollout buffer.  The step kernel then writes this step's observations THERE instead of the simulator's own tensor
        and the returned observations are views of ``out``: the per-step clone + insert of the reference's trainer
        (train/MAPPO/main_player.py:245-247, utils/shared_buffer.py:115) disappears, same bytes written.  ``static_observations``
        is not refreshed by such a step."""
        if out is not None:
            if out.shape != self.static_world_major_observations.shape:
                raise ValueError(f"out must have the world-major shape {tuple(self.static_world_major_observations.shape)}, got {tuple(out.shape)}")
            if out.data_ptr() != self._redirected_to:
                self.sim.set_observation_output(out)
                self._redirected_to = out.data_ptr()
        elif self._redirected_to:
            self.sim.set_observation_output(None)
            self._redirected_to = 0
        # (P, N, 1) int64/int32 on any device -> the simulator's int32 action tensor
        if (actions.dtype == torch.int64 and actions.device == self.static_actions.device and actions.is_contiguous() and
                actions.shape == self.static_actions.shape):
            # what the reference's harness passes (randint_like of a long tensor): the step kernel reads it as it is and
            # mirrors it into static_actions, which the reference fills with a copy kernel of its own (overcooked_env.py:107)
            self.sim._step_i64_checked(actions.data_ptr())
        else:
            self.static_actions.copy_(actions.to(self.static_actions.device), non_blocking=True)
            self.sim.step()
        if out is not None:
            obs = [VectorObservation(self.static_active_agents[i], out[:, i].transpose(1, 2), action_mask=self.static_action_mask) for i in range(self.num_players)]
            if not self._resident:
                obs = [VectorObservation(self.to_torch(o.active), self.to_torch(o.obs), action_mask=self.static_action_mask) for o in obs]
        else:
            obs = self.get_obs()
        if self._resident:
            return obs, self.static_rewards, self.static_dones, self.infos
        return obs, self.to_torch(self.static_rewards), self.to_torch(self.static_dones), self.infos

    def n_reset(self):
        """Like the reference (overcooked2_env.py:111-112) this restarts nothing: worlds restart themselves at
        the horizon inside ``step``."""
        return self.get_obs()

    def close(self, **kwargs):
        self.sim.close()
