#!/usr/bin/env python3
"""mrl_rollout_random for Cartpole by batch size: us per step as ONE persistent cooperative launch (mrl_cartpole_rollout) and as
one single-launch step per step with the action drawn in the kernel (cartpole.no_persistent)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode  # noqa: E402


def per_step(sim, steps):
    sim.rollout_random(50, seed=1, first_step=0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sim.rollout_random(steps, seed=1, first_step=50)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


for n in [int(a) for a in sys.argv[1:]] or [32, 1000, 10000, 100000, 262144, 524288, 1048576, 2097152]:
    row = []
    for knob in (0, 1):
        with _lib.debug_knobs({"cartpole.no_persistent": knob, "cartpole.persistent_max": 1 << 30}):
            sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
        row.append((sim.rollout_kernel_name, per_step(sim, 400)))
        sim.close()
    print(f"{n:8d} worlds: persistent {row[0][1]:7.2f} us ({row[0][0]}), one launch per step {row[1][1]:7.2f} us ({row[1][0]})", flush=True)
