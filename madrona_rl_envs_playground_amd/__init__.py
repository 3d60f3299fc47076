"""MI355X-native batched RL-environment step engine (Overcooked, Hanabi,
Cartpole) behind the API of willwng/madrona_rl_envs_playground.

    simulators            OvercookedSimulator / HanabiSimulator / CartpoleSimulator (+ madrona.ExecMode)
    envs                  OvercookedMadrona, HanabiMadrona, CartpoleMadronaTorch/Numpy
    pantheonrl_extension  VectorMultiAgentEnv, MadronaEnv, VectorObservation, VectorAgent
    layouts               Overcooked layout data + get_base_layout_params
    distributed           world sharding over the GPUs of a node, RCCL observation gather

All stepping is done by libmrl_envs.so (csrc/*.hip, gfx950) through the C ABI in
include/mrl_envs.h; importing a simulator without that library raises.
"""
__version__ = "0.1.0"
