"""``HanabiMadrona`` -- drop-in for /root/reference/envs/hanabi_env.py:13-104.
The reference sizes its spaces through DeepMind's hanabi_learning_environment;
here they come from ``hanabi_spec`` (same numbers, derived from the encoders)."""
import torch

from .. import hanabi_spec
from ..pantheonrl_extension.vectorenv import MadronaEnv
from ..simulators import ExecMode, HanabiSimulator
from ..spaces import Discrete, MultiBinary

DEFAULT_N = 2

FULL_CONFIG = {"colors": 5, "ranks": 5, "players": DEFAULT_N, "max_information_tokens": 8, "max_life_tokens": 3,
               "observation_type": 1}
SMALL_CONFIG = {"colors": 2, "ranks": 5, "players": DEFAULT_N, "hand_size": 2, "max_information_tokens": 3,
                "max_life_tokens": 1, "observation_type": 1}
VERY_SMALL_CONFIG = {"colors": 1, "ranks": 5, "players": DEFAULT_N, "hand_size": 5, "max_information_tokens": 3,
                     "max_life_tokens": 1, "observation_type": 1}
DEFAULT_CONFIG = VERY_SMALL_CONFIG
config_choice = {"very_small": VERY_SMALL_CONFIG, "small": SMALL_CONFIG, "full": FULL_CONFIG}


class HanabiMadrona(MadronaEnv):

    def __init__(self, num_envs, gpu_id, debug_compile=True, config=None, use_cpu=False, use_env_cpu=False):
        self.config = config if config is not None else DEFAULT_CONFIG
        config = self.config
        sim = HanabiSimulator(exec_mode=ExecMode.CPU if use_cpu else ExecMode.CUDA, gpu_id=gpu_id,
                              num_worlds=num_envs, colors=config["colors"], ranks=config["ranks"],
                              players=config["players"], max_information_tokens=config["max_information_tokens"],
                              max_life_tokens=config["max_life_tokens"], debug_compile=debug_compile)
        obs_size = hanabi_spec.observation_size(config)
        state_size = hanabi_spec.state_size(config)
        max_moves = hanabi_spec.num_moves(config)
        self.observation_space = MultiBinary(obs_size)
        self.action_space = Discrete(max_moves)
        self.share_observation_space = MultiBinary(state_size)
        device = torch.device("cpu") if use_env_cpu else None
        super().__init__(num_envs=num_envs, gpu_id=gpu_id, sim=sim, debug_compile=debug_compile, obs_size=obs_size,
                         state_size=state_size, discrete_action_size=max_moves, env_device=device)

    def close(self, **kwargs):
        self.sim.close()
