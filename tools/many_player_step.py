#!/usr/bin/env python3
"""us per step of the many_player_layout (15 x 17) with a given player count and batch:
`python tools/many_player_step.py <players> <worlds> [knob=value ...]` (the rows of the reference README's many-player table)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib, layouts
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator
players, n = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("="); _lib.debug_set(k, int(v))
params = layouts.get_base_layout_params("many_player_layout", 400, max_num_players=players)
P = params["num_players"]
sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(20): sim.step_with_actions(pool[i % 8])
torch.cuda.synchronize(); e0.record()
for i in range(500): sim.step_with_actions(pool[i % 8])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 2
print(f"{players} players {n} worlds {sys.argv[3:]}: {us:.2f} us per step, shape {sim.launch_shape}, {sim.kernel_name}")
