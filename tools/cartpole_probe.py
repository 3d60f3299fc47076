#!/usr/bin/env python3
"""Cartpole step at N worlds, one launch / two launches: us per step by HIP events and by the wall clock, stepping the
simulator's own ACTION tensor or a pool of caller tensors (what bench.py does).  Knobs: key=value arguments."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode  # noqa: E402


def measure(fn, steps=500, warmup=50):
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
    return e0.elapsed_time(e1) / steps * 1e3, (time.perf_counter() - t0) / steps * 1e6


def main():
    sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [1 << 20]
    knobs = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}
    for n in sizes:
        for fused in (1, 2):
            with _lib.debug_knobs({**knobs, "fused_step": fused}):
                sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
            sim.action_tensor().to_torch().copy_(torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda"))
            own = measure(lambda i: sim.step())
            rows = [f"own tensor {own[0]:6.2f} / {own[1]:6.2f}"]
            for k in (1, 2, 8):
                pool = [torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda") for _ in range(k)]
                r = measure(lambda i: sim.step_with_actions(pool[i % k]))
                rows.append(f"pool of {k} {r[0]:6.2f} / {r[1]:6.2f}")
            print(f"{n:8d} worlds {sim.kernel_name:24s} us per step (events / wall): " + "; ".join(rows), flush=True)
            sim.close()


if __name__ == "__main__":
    main()
