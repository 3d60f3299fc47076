#!/usr/bin/env python3
"""Long-run check of the persistent rollouts' grid-wide hand-offs: tens of thousands of steps in one-launch
chunks against the one-launch-per-step path, final tensors bit-identical, SCAN_TIMEOUT zero."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd._lib import debug_knobs  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, HanabiSimulator  # noqa: E402

FULL = dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3)


def pair(make, knob):
    a = make()
    with debug_knobs({knob: 1}):
        b = make()
    return a, b


def main():
    steps, chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 30000, 997
    a, b = pair(lambda: HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=65536, **FULL), "hanabi.no_persistent")
    done = 0
    while done < steps:
        k = min(chunk, steps - done)
        a.rollout_random(k, seed=5, first_step=done)
        b.rollout_random(k, seed=5, first_step=done)
        done += k
        for name in ("game_tensor", "agent_state_tensor", "action_mask_tensor", "active_agent_tensor", "reward_tensor", "done_tensor",
                     "action_tensor", "reset_count_tensor"):
            assert torch.equal(getattr(a, name)().to_torch(), getattr(b, name)().to_torch()), f"hanabi {name} differs after {done} steps"
    assert a.rollout_kernel_name == "mrl_hanabi_rollout" and b.rollout_kernel_name == b.kernel_name
    assert int(a.scan_timeout_tensor().to_torch().item()) == 0 and int(b.scan_timeout_tensor().to_torch().item()) == 0
    print("hanabi", done, "steps x 65536 worlds: persistent == per-step launches")
    a.close(); b.close()

    a, b = pair(lambda: CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=1 << 20), "cartpole.no_persistent")
    done = 0
    while done < steps:
        k = min(chunk, steps - done)
        a.rollout_random(k, seed=6, first_step=done)
        b.rollout_random(k, seed=6, first_step=done)
        done += k
        assert torch.equal(a.observation_tensor().to_torch(), b.observation_tensor().to_torch()), f"cartpole state differs after {done} steps"
    assert int(a.scan_timeout_tensor().to_torch().item()) == 0 and int(b.scan_timeout_tensor().to_torch().item()) == 0
    print("cartpole", done, "steps x 1M worlds: persistent == per-step launches")
    a.close(); b.close()


if __name__ == "__main__":
    main()
