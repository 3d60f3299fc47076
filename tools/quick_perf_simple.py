#!/usr/bin/env python3
"""Kernel-time triage for the Simplecooked step: `python tools/quick_perf_simple.py [layout] [worlds] [knob=value ...]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib, layouts  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, SimplecookedSimulator  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "simple"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    _lib.debug_set(k, int(v))
params = layouts.get_simplecooked_layout_params(layout, 400)
P = params["num_players"]
sim = SimplecookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda") for _ in range(16)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps):
    fn(0)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


step = timed(lambda i: sim.step_with_actions(pool[i % 16]), 2000)
roll = timed(lambda i: sim.rollout_random(500, seed=1, first_step=500 * i), 4) / 500
b = sim.bytes_per_world_step * n
print(f"{layout} {n} worlds [{' '.join(sys.argv[3:])}]: step {step:.2f} us ({b / step / 1e3:.0f} GB/s, {b / step / 8e6:.3f} of 8 TB/s)  "
      f"rollout {roll:.2f} us/step  kernel {sim.kernel_name}")
