#!/usr/bin/env python3
"""One launch (in-kernel look-back, csrc/episode_scan.hpp) or two per step for Cartpole / Hanabi, by batch size: us per step, call per step
from Python (what a small batch is bound by) -- the numbers behind the library's default (mrl_debug_set fused_step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, HanabiSimulator


def per_step(sim, reps):
    for _ in range(30):
        sim.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sim.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for game, sizes in (("cartpole", [32, 1000, 10000, 100000, 262144, 524288, 1048576]), ("hanabi", [32, 1000, 4096, 10000, 16384, 32768, 65536])):
    for n in sizes:
        row = []
        for knob in (1, 2):
            with _lib.debug_knobs({"fused_step": knob}):
                if game == "cartpole":
                    sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
                else:
                    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                                          max_information_tokens=8, max_life_tokens=3)
            if game == "cartpole":
                sim.action_tensor().to_torch().copy_(torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda"))
            row.append((sim.kernel_name, per_step(sim, 2000 if n <= 100000 else 500)))
            sim.close()
        print(f"{game} {n:8d} worlds: one launch {row[0][1]:7.2f} us ({row[0][0]}), two launches {row[1][1]:7.2f} us ({row[1][0]})", flush=True)
