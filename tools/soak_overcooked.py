#!/usr/bin/env python3
"""Long-run equality of the shipped Overcooked / Simplecooked kernels (specialised, direct encode, two groups per wave,
store flavour by the library's rules) with the generic searched-encode kernels: thousands of steps mixing per-step
launches on random actions, device-side rollouts and action sequences; every tensor compared after every chunk.
`python tools/soak_overcooked.py [steps]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import layouts  # noqa: E402
from madrona_rl_envs_playground_amd._lib import debug_knobs  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator, SimplecookedSimulator  # noqa: E402

GETS = ("observation_world_major_tensor", "reward_tensor", "done_tensor", "state_objects_tensor", "state_players_tensor",
        "state_timestep_tensor")


def soak(name, make, P, n, steps):
    shipped = make()
    with debug_knobs({"overcooked.no_fixed": 1, "overcooked.no_direct": 1, "overcooked.groups": 1, "overcooked.whole_store": 1}):
        plainest = make()
    gen = torch.Generator(device="cuda").manual_seed(7)
    done, chunk = 0, 0
    while done < steps:
        kind = chunk % 3
        if kind == 0:
            for _ in range(40):
                a = torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
                a[torch.rand((P, n, 1), device="cuda", generator=gen) < 0.25] = 5
                shipped.step_with_actions(a)
                plainest.step_with_actions(a)
            done += 40
        elif kind == 1:
            shipped.rollout_random(211, seed=3, first_step=done)
            plainest.rollout_random(211, seed=3, first_step=done)
            done += 211
        else:
            seq = torch.randint(0, 6, (64, P, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            shipped.step_sequence(seq)
            plainest.step_sequence(seq)
            done += 64
        chunk += 1
        for g in GETS:
            assert torch.equal(getattr(shipped, g)().to_torch(), getattr(plainest, g)().to_torch()), f"{name}: {g} differs after {done} steps"
    print(f"{name} {n} worlds, {done} steps: {shipped.kernel_name} == {plainest.kernel_name}", flush=True)
    shipped.close()
    plainest.close()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    for layout, n in (("cramped_room", 32768), ("asymmetric_advantages", 32768), ("coordination_ring", 40001), ("counter_circuit", 32768),
                      ("cramped_room", 131072)):
        params = layouts.get_base_layout_params(layout, 400)
        soak("overcooked " + layout, lambda: OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params), 2, n, steps)
    for layout, n in (("simple", 32768), ("unident_s", 32768), ("random0", 32771)):
        params = layouts.get_simplecooked_layout_params(layout, 400)
        soak("simplecooked " + layout, lambda: SimplecookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params), 2, n, steps)


if __name__ == "__main__":
    main()
