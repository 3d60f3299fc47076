"""Generates tests/golden/balance_transitions.npz from the REFERENCE's own one-step checker world for the
balance-beam game (envs/balance_beam_env.py:96-146, class PantheonLine, the numpy world its validate_step
:152-217 steps), imported from /root/reference in the build container (stub modules for the uninstalled gym /
build.* packages: ordinary ModuleNotFoundError, nothing was denied).  Run only there:

    python tests/golden/make_balance_golden.py

Data only: observation rows before a step (2, 7), the two actions, the rows after it, reward, done -- for
every state random play reaches (both agents' histories included).  Reset positions are not in it: the
reference draws them with numpy's global generator here and with its own rng.hpp in C++."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()
import types  # noqa: E402

sys.modules.setdefault("build.madrona_balance_example_python", types.ModuleType("build.madrona_balance_example_python"))
sys.modules["gym.spaces"].MultiDiscrete = sys.modules["gym.spaces"].Discrete
import envs.balance_beam_env as ref  # noqa: E402  (reference)


def rows(full_obs):
    return np.stack([np.asarray(full_obs[0][0], np.int64), np.asarray(full_obs[1][0], np.int64)]).astype(np.int32)


def main():
    np.random.seed(7)
    rng = np.random.default_rng(7)
    env = ref.PantheonLine()
    before, acts, after, rew, done = [], [], [], [], []
    for _ in range(4000):
        _, obs = env.n_reset()
        while True:
            a = rng.integers(0, 4, size=2)
            before.append(rows(obs))
            _, obs, r, d, _ = env.n_step([[int(a[0])], [int(a[1])]])
            acts.append(a.astype(np.int32))
            after.append(rows(obs))
            rew.append(np.float32(r[0]))
            done.append(int(d))
            if d:
                break
    out = os.path.join(HERE, "balance_transitions.npz")
    np.savez_compressed(out, before=np.stack(before), actions=np.stack(acts), after=np.stack(after),
                        reward=np.array(rew, np.float32), done=np.array(done, np.int32))
    b = np.stack(before)
    print(f"{len(before)} transitions, {len(np.unique(b.reshape(len(b), -1), axis=0))} distinct states, "
          f"{int(np.sum(done))} episode ends, rewards {sorted(set(np.round(rew, 3).tolist()))}, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
