"""Per-game env wrappers with the reference's class names
(/root/reference/envs/{overcooked,hanabi,cartpole}_env.py)."""
from .cartpole_env import CartpoleMadronaNumpy, CartpoleMadronaTorch  # noqa: F401
from .hanabi_env import FULL_CONFIG, SMALL_CONFIG, VERY_SMALL_CONFIG, HanabiMadrona, config_choice  # noqa: F401
from .overcooked_env import OvercookedMadrona, get_base_layout_params  # noqa: F401
