#!/usr/bin/env python3
"""Hanabi step timings for A/B runs of kernel variants on one box: µs per step of the single launch, the two-launch pair
and the persistent rollout (HIP events around back-to-back calls, the masked-random policy drawn on the device so that
the games stay in their steady state).  MRL_ENVS_LIB=<another build of the same sources> selects the variant."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator  # noqa: E402


def events(fn, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(steps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="+", default=[65536])
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--knob", action="append", default=[])
    args = ap.parse_args()
    out = {"lib": os.path.basename(_lib.LIB_PATH), "knobs": args.knob}
    for n in args.worlds:
        row = {}
        for name, knobs in (("one_launch", {"hanabi.no_persistent": 1}), ("two_launches", {"hanabi.no_persistent": 1, "fused_step": 2}),
                            ("rollout", {})):
            knobs = dict(knobs, **{k: int(v) for k, v in (kv.split("=") for kv in args.knob)})
            with _lib.debug_knobs(knobs):
                sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                                      max_information_tokens=8, max_life_tokens=3)
            sim.rollout_random(100, seed=3, first_step=0)  # into the steady state (games of every age)
            torch.cuda.synchronize()

            first = [100]

            def steps(k, sim=sim, first=first):  # one step call per launch, uniformly random legal moves drawn by the step kernel
                for _ in range(k):
                    sim.rollout_random(1, seed=3, first_step=first[0])
                    first[0] += 1

            def roll(k, sim=sim, first=first):  # all steps in one (persistent) launch
                sim.rollout_random(k, seed=3, first_step=first[0])
                first[0] += k
            fn = roll if name == "rollout" else steps
            fn(20)
            row[name + "_us"] = round(min(events(fn, args.steps) for _ in range(args.repeat)), 3)
            row[name + "_kernel"] = sim.rollout_kernel_name if name == "rollout" else sim.kernel_name
            sim.close()
        out[str(n)] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
