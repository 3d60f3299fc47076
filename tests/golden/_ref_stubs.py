"""Container-only helper for the golden-vector generators: stub modules for the
packages the reference imports but the image lacks (gym, overcooked_ai_py,
hanabi_learning_environment, tensorboard, the compiled build.* modules), so its
pure-Python oracles/checkers can be imported from /root/reference.  The missing
packages raise ordinary ModuleNotFoundError; nothing was denied."""
import sys
import types


def install():
    names = ["gym", "gym.spaces", "gym.vector", "gym.vector.vector_env", "gym.utils", "gym.utils.seeding",
             "gym.logger", "overcooked_ai_py", "overcooked_ai_py.utils", "overcooked_ai_py.mdp",
             "overcooked_ai_py.mdp.actions", "overcooked_ai_py.mdp.overcooked_mdp",
             "overcooked_ai_py.mdp.overcooked_env", "hanabi_learning_environment",
             "hanabi_learning_environment.rl_env", "build", "build.madrona_overcooked_example_python",
             "build.madrona_hanabi_example_python", "build.madrona_cartpole_example_python",
             "tensorboard", "torch.utils.tensorboard"]
    for name in names:
        sys.modules.setdefault(name, types.ModuleType(name))
    gym = sys.modules["gym"]
    gym.spaces = sys.modules["gym.spaces"]
    gym.logger = sys.modules["gym.logger"]
    gym.utils = sys.modules["gym.utils"]
    gym.Env = type("Env", (), {})
    sys.modules["gym.logger"].warn = lambda *a, **k: None
    sys.modules["gym.utils"].seeding = sys.modules["gym.utils.seeding"]
    import numpy as np
    sys.modules["gym.utils.seeding"].np_random = lambda seed=None: (np.random.default_rng(seed), seed)
    sys.modules["gym.vector.vector_env"].VectorEnv = type("VectorEnv", (), {})

    class _Space:
        def __init__(self, *a, **k):
            self.args = a

        def contains(self, x):
            return True

    for cls in ("Space", "Discrete", "MultiBinary", "Box", "MultiDiscrete"):
        setattr(sys.modules["gym.spaces"], cls, type(cls, (_Space,), {}))
    sys.modules["overcooked_ai_py.mdp.actions"].Action = type("Action", (), {"NUM_ACTIONS": 6})
    sys.modules["overcooked_ai_py.mdp.overcooked_mdp"].OvercookedGridworld = object
    sys.modules["overcooked_ai_py.mdp.overcooked_env"].OvercookedEnv = object
    sys.modules["overcooked_ai_py.utils"].read_layout_dict = lambda name: None
    sys.modules["overcooked_ai_py.utils"].load_dict_from_file = lambda path: None
    sys.modules["hanabi_learning_environment.rl_env"].HanabiEnv = object
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    if "/root/reference" not in sys.path:
        sys.path.insert(1, "/root/reference")
