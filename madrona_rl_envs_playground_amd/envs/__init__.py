"""Per-game env wrappers with the reference's class names
(/root/reference/envs/{overcooked,overcooked2,hanabi,cartpole,balance_beam}_env.py).  ``overcooked2_env`` also
defines a class called ``OvercookedMadrona`` (as in the reference): import it from its module."""
from .balance_beam_env import BalanceMadronaTorch  # noqa: F401
from .cartpole_env import CartpoleMadronaNumpy, CartpoleMadronaTorch  # noqa: F401
from .hanabi_env import FULL_CONFIG, SMALL_CONFIG, VERY_SMALL_CONFIG, HanabiMadrona, config_choice  # noqa: F401
from .overcooked_env import OvercookedMadrona, get_base_layout_params  # noqa: F401
