#!/usr/bin/env python3
"""Minimal driver for rocprofv3 passes: N steps of one game, nothing else.
    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_step.py --game overcooked --worlds 32768 --steps 200
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="overcooked", choices=["overcooked", "hanabi", "cartpole", "balance", "simplecooked"])
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--worlds", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--players", type=int, default=None)
    args = ap.parse_args()
    from madrona_rl_envs_playground_amd import layouts
    from madrona_rl_envs_playground_amd.simulators import (CartpoleSimulator, ExecMode, HanabiSimulator,
                                                           OvercookedSimulator)
    n = args.worlds
    if args.game == "overcooked":
        params = layouts.get_base_layout_params(args.layout, 400, max_num_players=args.players)
        sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
        pool = [torch.randint(0, 6, (params["num_players"], n, 1), dtype=torch.int32, device="cuda") for _ in range(16)]
        for i in range(args.steps):
            sim.step_with_actions(pool[i % 16])
    elif args.game == "balance":
        from madrona_rl_envs_playground_amd.simulators import BalanceBeamSimulator
        sim = BalanceBeamSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
        pool = [torch.randint(0, 4, (2, n, 1), dtype=torch.int32, device="cuda") for _ in range(16)]
        for i in range(args.steps):
            sim.step_with_actions(pool[i % 16])
    elif args.game == "simplecooked":
        from madrona_rl_envs_playground_amd.simulators import SimplecookedSimulator
        params = layouts.get_simplecooked_layout_params("simple", 400)
        sim = SimplecookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
        pool = [torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda") for _ in range(16)]
        for i in range(args.steps):
            sim.step_with_actions(pool[i % 16])
    elif args.game == "cartpole":
        sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
        pool = [torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda") for _ in range(16)]
        for i in range(args.steps):
            sim.step_with_actions(pool[i % 16])
    else:
        from madrona_rl_envs_playground_amd import _lib
        with _lib.debug_knobs({"hanabi.no_persistent": 1}):  # (no cooperative launch under the profiler: rocprofv3 --pmc crashed at exit behind one)
            sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                                  max_information_tokens=8, max_life_tokens=3)
        mask = sim.action_mask_tensor().to_torch()
        act = sim.action_tensor().to_torch()
        # into the steady state first (games of every age, 7.8 % of the worlds finishing per step under the random policy): the
        # first steps of 65536 fresh games finish nobody, and a finished world costs two blocks instead of one.  These are
        # launches of the same step kernel: tools/pmc_traffic.py averages the LAST `--steps` dispatches only.
        sim.rollout_random(200, seed=1, first_step=0)
        for i in range(args.steps):
            act.copy_((torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True).to(torch.int32))
            sim.step()
    torch.cuda.synchronize()
    print("done", args.game, n, args.steps, sim.kernel_name, sim.bytes_per_world_step)


if __name__ == "__main__":
    main()
