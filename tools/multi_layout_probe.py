#!/usr/bin/env python3
"""The five standard Overcooked layouts as one batch, us per step (all five sub-batches): five separate envs stepped one after
the other, OvercookedMultiLayout on forked streams, as one captured graph, and as ONE launch (mrl_step_many)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd.envs import OvercookedMadrona  # noqa: E402
from madrona_rl_envs_playground_amd.envs.multi_layout import OvercookedMultiLayout  # noqa: E402

STANDARD = ["cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit"]


def per_step(fn, reps=1500):
    for i in range(30):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


out = {}
for n in (100, 1000, 8192, 32768):
    acts = [[torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda") for _ in STANDARD] for _ in range(8)]
    row = {}
    singles = [OvercookedMadrona(name, n, 0) for name in STANDARD]

    def one_after_the_other(i):
        for env, a in zip(singles, acts[i % 8]):
            env.sim.step_with_actions(a)
    row["five_step_calls"] = per_step(one_after_the_other)
    for env in singles:
        env.close()
    for mode in ("sequential", "forked_streams", "graph", "one_launch"):
        multi = OvercookedMultiLayout(STANDARD, n, 0, mode=mode)
        row[mode] = per_step(lambda i: multi.n_step(acts[i % 8]), reps=600 if n >= 8192 else 1500)
        multi.close()
    out[n] = row
print(json.dumps(out))
