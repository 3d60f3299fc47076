#!/usr/bin/env python3
"""The two scaling tables of the reference's README (src/overcooked_env/README.org:101-121) on this engine:
cramped_room by number of environments, many_player_layout (15x17) by number of players at 1000 environments.
env-steps/s = worlds / wall time per step, random actions pre-sampled on the device, back-to-back
`step_with_actions` launches from Python (so small batches are bound by the ~5 us per launch a Python caller
can issue, not by the GPU)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import layouts  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator  # noqa: E402


def rate(layout, n, players=None, steps=600):
    params = layouts.get_base_layout_params(layout, 400, max_num_players=players)
    P = params["num_players"]
    sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
    pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
    for i in range(20):
        sim.step_with_actions(pool[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        sim.step_with_actions(pool[i % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sim.rollout_random(10, seed=1, first_step=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.rollout_random(steps, seed=1, first_step=10)
    torch.cuda.synchronize()
    dr = (time.perf_counter() - t0) / steps
    sim.close()
    return {"worlds": n, "players": P, "us_per_step": dt * 1e6, "steps_per_s": n / dt, "device_policy_steps_per_s": n / dr}


def main():
    out = {"cramped_room_by_envs": [rate("cramped_room", n) for n in (32, 100, 1000, 10000, 100000)],
           "many_player_layout_1000_envs_by_players": [rate("many_player_layout", 1000, players=p, steps=300) for p in (2, 4, 8, 16, 30)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
