"""``BalanceMadronaTorch`` -- drop-in for /root/reference/envs/balance_beam_env.py:20-40: the generic
``MadronaEnv`` wrapper over ``BalanceBeamSimulator`` with the reference's spaces (an observation of six
positions in 0..8 -- own and partner history, shifted by BUFFER -- and the steps left)."""
import torch

from ..pantheonrl_extension.vectorenv import MadronaEnv
from ..simulators import BalanceBeamSimulator, ExecMode
from ..spaces import Discrete, MultiDiscrete

NUM_SPACES = 5
VALID_MOVES = [-2, -1, 1, 2]
BUFFER = 2
TIME = 3
SCALE = 0.2


class BalanceMadronaTorch(MadronaEnv):

    def __init__(self, num_envs, gpu_id, debug_compile=True, use_cpu=False, use_env_cpu=False):
        sim = BalanceBeamSimulator(exec_mode=ExecMode.CPU if use_cpu else ExecMode.CUDA, gpu_id=gpu_id,
                                   num_worlds=num_envs, debug_compile=debug_compile)
        device = torch.device("cpu") if use_env_cpu else None
        super().__init__(num_envs, gpu_id, sim, env_device=device)
        self.observation_space = MultiDiscrete([NUM_SPACES + 2 * BUFFER] * 2 * TIME + [TIME])
        self.action_space = Discrete(len(VALID_MOVES))
        self.share_observation_space = self.observation_space

    def close(self, **kwargs):
        self.sim.close()
