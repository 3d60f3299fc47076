#!/usr/bin/env python3
"""The four arithmetic variants of the Cartpole transition (csrc/cartpole.hip, mrl_debug_set cartpole.variant) side by side:
accuracy against the reference-generated float64 transitions (tests/golden/cartpole_transitions.npz; the reference's own
bound is 1e-6, envs/cartpole_env.py:277), how many one-step results differ in any bit from the reference-typed variant on a
random-policy state distribution, and us per step at 1 M worlds as two launches, one launch and the persistent rollout."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cartpole_transitions.npz")
NAMES = {1: "reference-typed", 2: "lean f64", 3: "lean f64 + bounded sincos", 4: "float"}


def make(n, **knobs):
    with _lib.debug_knobs(knobs):
        return CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)


def us_per_step(fn, steps=400, warmup=30):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    z = np.load(GOLDEN)
    states, actions, next64 = z["states"], z["actions"], z["next64"]
    m = len(states)
    # a random-policy state distribution: 1 M worlds after 37 steps of the reference-typed variant
    ref = make(n, **{"cartpole.variant": 1})
    ref.rollout_random(37, seed=5, first_step=0)
    start = ref.observation_tensor().to_torch().clone()
    act = torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda")
    ref.step_phase1(act)
    ref_next = ref.observation_tensor().to_torch().clone()
    ref_done = ref.reset_tensor().to_torch().clone()
    ref.close()
    out = {"worlds": n, "variants": {}}
    for v in (1, 2, 3, 4):
        row = {"name": NAMES[v]}
        g = make(m, **{"cartpole.variant": v})
        g.observation_tensor().to_torch().copy_(torch.from_numpy(states).cuda())
        g.action_tensor().to_torch().copy_(torch.from_numpy(actions).cuda().view(m, 1))
        g.step_phase1(None)
        got = g.observation_tensor().to_torch().cpu().numpy().astype(np.float64)
        row["golden_max_abs_err"] = float(np.abs(got - next64).max())
        row["golden_done_disagree"] = int((g.reset_tensor().to_torch().cpu().numpy()[:, 0] != z["done"]).sum())
        g.close()
        s = make(n, **{"cartpole.variant": v})
        s.observation_tensor().to_torch().copy_(start)
        s.step_phase1(act)
        nxt = s.observation_tensor().to_torch()
        row["worlds_differing_from_reference_typed"] = int((nxt.view(torch.int32) != ref_next.view(torch.int32)).any(dim=1).sum())
        row["max_abs_diff_from_reference_typed"] = float((nxt.double() - ref_next.double()).abs().max())
        row["done_flags_differing"] = int((s.reset_tensor().to_torch() != ref_done).sum())
        s.close()
        pool = [torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
        for label, knob in (("two_launches_us", 2), ("one_launch_us", 1)):
            s = make(n, **{"cartpole.variant": v, "fused_step": knob})
            row[label] = us_per_step(lambda i: s.step_with_actions(pool[i % 8]))
            row[label + "_kernel"] = s.kernel_name
            s.close()
        s = make(n, **{"cartpole.variant": v})
        s.rollout_random(20, seed=1, first_step=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s.rollout_random(400, seed=1, first_step=20)
        e1.record()
        torch.cuda.synchronize()
        row["persistent_rollout_us"] = e0.elapsed_time(e1) / 400 * 1e3
        row["rollout_kernel"] = s.rollout_kernel_name
        s.close()
        out["variants"][str(v)] = row
        print(json.dumps({str(v): row}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
