"""``CartpoleMadronaTorch`` / ``CartpoleMadronaNumpy`` -- drop-ins for
/root/reference/envs/cartpole_env.py:27-128 (gym.vector.VectorEnv-style:
``step(actions) -> (obs, rewards, dones, infos)``, ``reset()`` returns the
current observations without restarting anything, :124-125)."""
from math import pi

import numpy as np
import torch

from ..simulators import CartpoleSimulator, ExecMode
from ..spaces import Box, Discrete

X_THRESHOLD = 2.4
THETA_THRESHOLD_RADIANS = 12 * 2 * pi / 360


class _CartpoleBase:
    def __init__(self, num_envs, gpu_id, debug_compile=True, use_cpu=False, use_env_cpu=False):
        high = np.array([X_THRESHOLD * 2, np.finfo(np.float32).max, THETA_THRESHOLD_RADIANS * 2,
                         np.finfo(np.float32).max], dtype=np.float32)
        self.num_envs = num_envs
        self.single_action_space = self.action_space = Discrete(2)
        self.single_observation_space = self.observation_space = Box(-high, high, dtype=np.float32)
        self.sim = CartpoleSimulator(exec_mode=ExecMode.CPU if use_cpu else ExecMode.CUDA, gpu_id=gpu_id,
                                     num_worlds=num_envs, debug_compile=debug_compile)
        self.static_dones = self.sim.reset_tensor().to_torch()
        self.static_actions = self.sim.action_tensor().to_torch()
        self.static_observations = self.sim.observation_tensor().to_torch()
        self.static_rewards = self.sim.reward_tensor().to_torch()
        self.device = torch.device("cpu") if use_env_cpu else self.static_observations.device
        self.infos = [{}] * self.num_envs

    def close(self, **kwargs):
        self.sim.close()


class CartpoleMadronaTorch(_CartpoleBase):
    def to_torch(self, a):
        return a.to(self.device)

    def step(self, actions):
        self.static_actions.copy_(actions[:, None].to(self.static_actions.device), non_blocking=True)
        self.sim.step()
        return (self.to_torch(self.static_observations), self.to_torch(self.static_rewards),
                self.to_torch(self.static_dones[:, 0]), self.infos)

    def reset(self):
        return self.to_torch(self.static_observations)


class CartpoleMadronaNumpy(_CartpoleBase):
    def step(self, actions):
        self.static_actions.copy_(torch.from_numpy(np.asarray(actions)[:, np.newaxis]))
        self.sim.step()
        return (self.static_observations.cpu().numpy(), self.static_rewards.cpu().numpy(),
                self.static_dones[:, 0].cpu().numpy(), [{}] * self.num_envs)

    def reset(self):
        return self.static_observations.cpu().numpy()
