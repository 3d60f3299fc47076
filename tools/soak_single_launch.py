#!/usr/bin/env python3
"""Long-run check of the single-launch step (self-healing look-back, scan wave, leader waves) against the two-launch pair:
tens of thousands of steps under the device-side random policy, one launch per step, three simulators side by side --
the single launch, the single launch with workgroups made to arrive late (`fused_heal_test`: the recount path runs all
the time), and the two-launch pair -- every tensor compared after every chunk."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd._lib import debug_knobs  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import BalanceBeamSimulator, CartpoleSimulator, ExecMode, HanabiSimulator  # noqa: E402

FULL = dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3)


def trio(make, extra):
    sims = []
    for knobs in ({"fused_step": 1}, {"fused_step": 1, "fused_heal_test": 7}, {"fused_step": 2}):
        with debug_knobs(dict(knobs, **extra)):
            sims.append(make())
    return sims


def run(name, make, extra, getters, steps, chunk, late_every):
    sims = trio(make, extra)
    print(name, [s.kernel_name for s in sims], flush=True)
    done = 0
    while done < steps:
        k = min(chunk, steps - done)
        sims[0].rollout_random(k, seed=5, first_step=done)
        sims[2].rollout_random(k, seed=5, first_step=done)
        # the late-workgroup variant waits out a ~100 us poll budget per recount: give it a slice of every chunk
        sims[1].rollout_random(k, seed=5, first_step=done)
        done += k
        for g in getters:
            ref = getattr(sims[2], g)().to_torch()
            for which, s in (("single launch", sims[0]), ("single launch, late workgroups", sims[1])):
                assert torch.equal(ref, getattr(s, g)().to_torch()), f"{name}: {g} of the {which} differs from the two-launch pair after {done} steps"
        if done % (chunk * late_every) == 0:
            print(f"  {done} steps equal", flush=True)
    assert all(int(s.scan_timeout_tensor().to_torch().item()) == 0 for s in sims if hasattr(s, "scan_timeout_tensor"))
    print(f"{name}: {done} steps: single launch == single launch with late workgroups == two launches")
    for s in sims:
        s.close()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    hanabi = ["game_tensor", "observation_tensor", "agent_state_tensor", "action_mask_tensor", "reward_tensor", "done_tensor",
              "active_agent_tensor", "reset_count_tensor", "action_tensor"]
    for n in (65536, 70001):
        run(f"hanabi {n} worlds", lambda: HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **FULL),
            {"hanabi.no_persistent": 1}, hanabi, steps if n == 65536 else steps // 4, 499, 8)
    cartpole = ["observation_tensor", "reset_tensor", "reward_tensor", "reset_count_tensor", "action_tensor"]
    for n in (1 << 20, 300001):
        run(f"cartpole {n} worlds", lambda: CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n),
            {"cartpole.no_persistent": 1}, cartpole, steps if n == 1 << 20 else steps // 4, 499, 8)
    balance = ["observation_tensor", "done_tensor", "reward_tensor", "reset_count_tensor", "action_tensor"]
    for n in (1 << 20, 300001):
        run(f"balance beam {n} worlds", lambda: BalanceBeamSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n),
            {}, balance, steps // 2 if n == 1 << 20 else steps // 4, 499, 8)


if __name__ == "__main__":
    main()
