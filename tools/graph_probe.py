#!/usr/bin/env python3
"""How much of a step's wall time is the host's launch rate?  The same back-to-back steps issued (a) one C-ABI call per
step from Python and (b) as a captured HIP graph of `--per-graph` launches replayed from the host (no per-launch host work).
With the diagnostic build (MRL_ENVS_LIB=.../libmrl_envs_diag.so) `--ablate` masks phases (16 = empty launch, 32 = loads
only, 8 = no encode, 2 = encode without the HBM stores); with the shipped library it must stay 0.
`python tools/graph_probe.py [--layout L] [--worlds N] [--ablate a,b,...]`."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib, layouts  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layout", default="cramped_room")
ap.add_argument("--worlds", type=int, default=32768)
ap.add_argument("--ablate", default="0")
ap.add_argument("--per-graph", type=int, default=200)
ap.add_argument("--knob", action="append", default=[])
args = ap.parse_args()
params = layouts.get_base_layout_params(args.layout, 400)
P, n = params["num_players"], args.worlds
pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda") for _ in range(16)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for ab in [int(x) for x in args.ablate.split(",")]:
    knobs = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.knob}
    if ab:
        knobs["ablate"] = ab
    with _lib.debug_knobs(knobs):  # cleared again even if the create throws: the knobs are process-global
        sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
    for i in range(50):
        sim.step_with_actions(pool[i % 16])
    torch.cuda.synchronize()
    e0.record()
    for i in range(2000):
        sim.step_with_actions(pool[i % 16])
    e1.record()
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) * 1e3 / 2000
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(args.per_graph):
                sim.step_with_actions(pool[i % 16])
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    graph = e0.elapsed_time(e1) * 1e3 / (10 * args.per_graph)
    print(f"{args.layout} {n} worlds ablate={ab} {args.knob}: call per step {eager:.2f} us, graph replay {graph:.2f} us per step  ({sim.kernel_name})", flush=True)
    sim.close()
