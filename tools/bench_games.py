#!/usr/bin/env python3
"""Throughput of the Hanabi and Cartpole steps (parity-test configs of BASELINE.json, not the
headline): env-steps/s with actions resident in HBM, back-to-back launches, HIP events."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import layouts  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import (BalanceBeamSimulator, CartpoleSimulator, ExecMode, HanabiSimulator,  # noqa: E402
                                                       SimplecookedSimulator)


def timed(fn, steps, warmup=20):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, e0.elapsed_time(e1) / steps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hanabi-worlds", type=int, default=65536)
    ap.add_argument("--cartpole-worlds", type=int, default=1048576)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--knob", action="append", default=[], help="key=value for mrl_debug_set (measurement variants)")
    args = ap.parse_args()
    from madrona_rl_envs_playground_amd import _lib
    for kv in args.knob:
        k, v = kv.split("=")
        _lib.debug_set(k, int(v))

    n = args.hanabi_worlds
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                          max_information_tokens=8, max_life_tokens=3)
    mask, act = sim.action_mask_tensor().to_torch(), sim.action_tensor().to_torch()

    def hanabi_masked(i):  # the reference harness: argmax(rand * mask) of each agent's current mask
        act.copy_((torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True))
        sim.step()

    wall, dev = timed(hanabi_masked, args.steps)
    out = {"hanabi": {"worlds": n, "steps_per_s_with_sampling": n / wall, "us_per_step_with_sampling": wall * 1e6}}
    # step alone: replay a recorded legal action stream is not possible without the sampling, so time
    # the two kernels with the last sampled actions kept (illegal moves are memory-safe, timing only)
    wall, dev = timed(lambda i: sim.step(), args.steps)
    out["hanabi"].update({"us_per_step_kernels_only": dev * 1e6, "steps_per_s_kernels_only": n / dev,
                          "algorithmic_GBps": sim.bytes_per_world_step * n / dev / 1e9})
    sim.rollout_random(20, seed=1, first_step=0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sim.rollout_random(args.steps, seed=1, first_step=20)
    e1.record()
    torch.cuda.synchronize()
    dev = e0.elapsed_time(e1) / args.steps * 1e-3
    persistent = sim.rollout_kernel_name == "mrl_hanabi_rollout"
    roll_bytes = sim.bytes_per_world_step - (2 * 176 if persistent else 0)  # (the records stay in LDS between the steps)
    out["hanabi"].update({"us_per_step_device_policy": dev * 1e6, "steps_per_s_device_policy": n / dev, "rollout_kernel": sim.rollout_kernel_name,
                          "device_policy_bytes_per_world_step": roll_bytes, "algorithmic_GBps_device_policy": roll_bytes * n / dev / 1e9})
    sim.close()

    n = args.cartpole_worlds
    sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
    pool = [torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
    wall, dev = timed(lambda i: sim.step_with_actions(pool[i % 8]), args.steps)
    out["cartpole"] = {"worlds": n, "us_per_step": dev * 1e6, "steps_per_s": n / dev,
                       "algorithmic_GBps": sim.bytes_per_world_step * n / dev / 1e9}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim.rollout_random(20, seed=1, first_step=0)
    torch.cuda.synchronize()
    e0.record()
    sim.rollout_random(args.steps, seed=1, first_step=20)
    e1.record()
    torch.cuda.synchronize()
    dev = e0.elapsed_time(e1) / args.steps * 1e-3
    out["cartpole"].update({"us_per_step_device_policy": dev * 1e6, "steps_per_s_device_policy": n / dev})
    sim.close()
    # the sibling worlds: Simplecooked `simple` at the headline's batch size, balance beam
    n = 32768
    params = layouts.get_simplecooked_layout_params("simple", 200)
    sim = SimplecookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
    pool = [torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
    wall, dev = timed(lambda i: sim.step_with_actions(pool[i % 8]), 1000)
    gbps = sim.bytes_per_world_step * n / dev / 1e9
    out["simplecooked"] = {"layout": "simple", "worlds": n, "us_per_step": dev * 1e6, "steps_per_s": n / dev, "bytes_per_world_step":
                           sim.bytes_per_world_step, "algorithmic_GBps": gbps, "frac_of_8TBps": gbps / 8000.0, "kernel": sim.kernel_name}
    wall, dev = timed(lambda i: sim.rollout_random(500, seed=3, first_step=500 * i), 4)
    roll_bytes = 2 * 20 * 20 + 8 + 4  # observations + rewards + done flag; state and actions stay on the chip
    out["simplecooked"].update({"us_per_step_device_policy": dev / 500 * 1e6, "steps_per_s_device_policy": n * 500 / dev,
                                "device_policy_bytes_per_world_step": roll_bytes,
                                "algorithmic_GBps_device_policy": roll_bytes * n * 500 / dev / 1e9})
    sim.close()
    n = 1 << 20
    sim = BalanceBeamSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
    pool = [torch.randint(0, 4, (2, n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
    wall, dev = timed(lambda i: sim.step_with_actions(pool[i % 8]), args.steps)
    out["balance_beam"] = {"worlds": n, "us_per_step": dev * 1e6, "steps_per_s": n / dev,
                           "algorithmic_GBps": sim.bytes_per_world_step * n / dev / 1e9}
    sim.close()
    out["knobs"] = args.knob
    print(json.dumps(out))


if __name__ == "__main__":
    main()
