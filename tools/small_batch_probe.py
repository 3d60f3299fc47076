#!/usr/bin/env python3
"""Small batches (the reference README's 32 / 100 / 1000 environments, src/overcooked_env/README.org:100-106) are bound by
the host's cost per step call, not by the GPU.  us per step for: the wrapped call (env.n_step), the bare C-ABI call, a
captured HIP graph of ONE step replayed per step, a graph of 16 steps replayed, and mrl_step_sequence (16 steps per call)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd.envs import OvercookedMadrona  # noqa: E402


def per_call(fn, reps=3000):
    for i in range(50):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


out = {}
for n in (32, 100, 1000, 10000):
    env = OvercookedMadrona("cramped_room", n, 0, horizon=400)
    sim = env.sim
    a64 = [torch.randint(0, 6, (2, n, 1), device="cuda") for _ in range(16)]
    a32 = [a.to(torch.int32) for a in a64]
    row = {"env_n_step": per_call(lambda i: env.n_step(a64[i % 16])), "c_abi_step_with_actions": per_call(lambda i: sim.step_with_actions(a32[i % 16]))}
    for k in (1, 16):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for j in range(k):
                    sim.step_with_actions(a32[j])
        row[f"graph_of_{k}_replayed_per_step"] = per_call(lambda i: g.replay(), reps=3000 // k) / k
        del g
    seq = torch.stack(a32).contiguous()
    row["step_sequence_16_per_step"] = per_call(lambda i: sim.step_sequence(seq), reps=300) / 16
    out[n] = row
    env.close()
print(json.dumps(out))
