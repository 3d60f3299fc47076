#!/usr/bin/env python3
"""Host-side cost of one step call (small batches are bound by it): microseconds per call of the pieces, measured with a
32-world simulator so that the GPU is never the bottleneck."""
import os
import sys
import time

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import layouts
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator
from madrona_rl_envs_playground_amd.envs import OvercookedMadrona

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
params = layouts.get_base_layout_params("cramped_room", 400)
sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
a = torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda")
a64 = a.to(torch.int64)
K = 20000


def per_call(fn):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / K * 1e6


L = sim._L
h = sim._handle
ptr = a.data_ptr()
stream = torch.cuda.current_stream(0).cuda_stream
print(f"worlds {n}")
print(f"  empty python call                      {per_call(lambda: None):6.2f} us")
print(f"  torch.cuda.current_stream().cuda_stream {per_call(lambda: torch.cuda.current_stream(0).cuda_stream):6.2f} us")
print(f"  action validation (_action_pointer)     {per_call(lambda: sim._action_pointer(a)):6.2f} us")
print(f"  ctypes mrl_step_with_actions alone      {per_call(lambda: L.mrl_step_with_actions(h, ptr, stream)):6.2f} us")
print(f"  sim.step_with_actions                   {per_call(lambda: sim.step_with_actions(a)):6.2f} us")
print(f"  sim.step_with_actions_i64               {per_call(lambda: sim.step_with_actions_i64(a64)):6.2f} us")
print(f"  sim.step (ACTION tensor)                {per_call(lambda: sim.step()):6.2f} us")
env = OvercookedMadrona("cramped_room", num_envs=n, gpu_id=0)
acts = torch.randint(0, 6, (2, n, 1), dtype=torch.int64, device="cuda")
print(f"  OvercookedMadrona.n_step                {per_call(lambda: env.n_step(acts)):6.2f} us")
