#!/usr/bin/env python3
"""Where the Hanabi step's HBM bytes above the algorithmic count come from: the average number of worlds
that finish per step under the masked-random policy -- each costs both agents' 896-byte blocks anew on top of the mover's --
next to the fixed extras (32 bytes of padding + one unused byte per block, status words, the ACTION tensor under the device
policy).  Compare with profiles/step_traffic.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
with _lib.debug_knobs({"hanabi.no_persistent": 1}):
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3)
sim.rollout_random(300, seed=11, first_step=0)
count = sim.reset_count_tensor().to_torch()
total = 0
steps = 300
for i in range(steps):
    sim.rollout_random(1, seed=11, first_step=300 + i)
    total += int(count.item())
per_step = total / steps
rows = 896
algorithmic = sim.bytes_per_world_step * n
out = {
    "worlds": n, "finished_worlds_per_step": per_step, "share": per_step / n,
    "algorithmic_bytes": algorithmic,
    "writes_expected": {
        "mover_block": n * rows, "records": n * 176, "active_reward_done": n * 20,
        "both_blocks_of_finished_worlds": per_step * 2 * rows, "action_tensor_device_policy": n * 4,
    },
    "reads_expected": {"records": n * 176, "actions": n * 8},
    "padding_and_unused_bytes_in_mover_block": n * (rows - 783 - 80),
}
out["writes_expected_total"] = sum(out["writes_expected"].values())
out["reads_expected_total"] = sum(out["reads_expected"].values())
print(json.dumps(out))
sim.close()
