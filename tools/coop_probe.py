#!/usr/bin/env python3
"""What one cooperative launch costs: mrl_rollout_random(1 step) per call (one hipLaunchCooperativeKernel per step,
records loaded and stored every call) against one call of many steps and against mrl_step (two launches)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, HanabiSimulator  # noqa: E402


def us(fn, reps):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i + 1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out = {}
for name, make, n in (("hanabi", lambda k: HanabiSimulator(ExecMode.CUDA, 0, k, 5, 5, 2, 8, 3), 65536),
                      ("cartpole", lambda k: CartpoleSimulator(ExecMode.CUDA, 0, k), 1 << 20)):
    sim = make(n)
    sim.rollout_random(50, seed=1, first_step=0)
    one = us(lambda i: sim.rollout_random(1, seed=1, first_step=50 + i), 300)
    many = us(lambda i: sim.rollout_random(300, seed=1, first_step=400 + 300 * i), 2) / 300
    two = us(lambda i: sim.step(), 300)
    out[name] = {"worlds": n, "us_per_step_one_cooperative_launch_per_step": one, "us_per_step_persistent_300": many,
                 "us_per_step_mrl_step": two}
    sim.close()
print(json.dumps(out))
