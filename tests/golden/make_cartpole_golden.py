"""Generates tests/golden/cartpole_transitions.npz with the REFERENCE's float64
Cartpole (envs/cartpole_env.py:130-233, class CartpoleNumpy), imported from
/root/reference in the build container (gym stubbed, see _ref_stubs.py).

Fixture = data only: float32 start states, actions, and the next state / done
the reference computes for each -- the same one-step construction its own
validation uses (envs/cartpole_env.py:246-288, tolerance 1e-6).  States are fed
as Python floats (exact values of the float32 inputs) so the arithmetic is
float64 throughout, as under the NumPy 1.x the reference targets (NumPy 2 would
keep float32 scalars in float32).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()
from envs.cartpole_env import CartpoleNumpy  # noqa: E402  (reference)


def main():
    rng = np.random.default_rng(0)
    m = 6000
    states = np.empty((m, 4), np.float32)
    states[:, 0] = rng.uniform(-2.6, 2.6, m)
    states[:, 1] = rng.normal(0, 1.5, m)
    states[:, 2] = rng.uniform(-0.25, 0.25, m)
    states[:, 3] = rng.normal(0, 2.0, m)
    # a band right at the thresholds and the reset range
    states[:500, 0] = np.float32(2.4) + rng.uniform(-2e-3, 2e-3, 500).astype(np.float32)
    states[500:1000, 2] = np.float32(12 * 2 * np.pi / 360) + rng.uniform(-2e-3, 2e-3, 500).astype(np.float32)
    states[1000:2500] = rng.uniform(-0.05, 0.05, (1500, 4))
    actions = rng.integers(0, 2, m).astype(np.int32)
    env = CartpoleNumpy()
    nxt64 = np.empty((m, 4), np.float64)
    done = np.empty((m,), np.int32)
    for i in range(m):
        env.steps_beyond_done = None
        env.state = tuple(float(v) for v in states[i])
        _, _, d, _ = env.step(int(actions[i]))
        nxt64[i] = env.state
        done[i] = int(d)
    out = os.path.join(HERE, "cartpole_transitions.npz")
    np.savez_compressed(out, states=states, actions=actions, next64=nxt64, done=done)
    print(f"{m} transitions, {int(done.sum())} terminal, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
