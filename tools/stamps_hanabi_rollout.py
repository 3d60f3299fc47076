"""Per-wave timeline of one step (the last but one) of the persistent Hanabi rollout (diagnostic build: make -C
madrona_rl_envs_playground_amd/csrc diag).  Stamps are s_memrealtime (100 MHz): microseconds after the step's earliest stamp."""
import os, sys, torch, numpy as np
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import _lib
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
with _lib.debug_knobs({"stamps": 1}):
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                          max_information_tokens=8, max_life_tokens=3)
assert sim.rollout_kernel_name == "mrl_hanabi_rollout", sim.rollout_kernel_name
sim.rollout_random(100, seed=3, first_step=0)
torch.cuda.synchronize()
st = sim._tensor(14).to_torch().cpu().numpy().view(np.uint64).reshape(-1, 8, 16).astype(np.float64) / 100.0  # (block, wave, stamp) in us
def row(name, v):
    v = v[np.isfinite(v)]
    print(f"{name:>44s}  {np.percentile(v,10):6.2f} {np.median(v):6.2f} {np.percentile(v,90):6.2f} {v.max():6.2f}")
a, bw, scan = st[:, 0:4, 0:4], st[:, :, 4:8], st[:, 0, 8:12]
t0 = a[:, :, 0].min()
print(f"{st.shape[0]} workgroups; us after the earliest phase-A wave reached the step: p10 / p50 / p90 / max")
for k, name in enumerate(["A: reached the step", "A: previous step's games dealt", "A: buffer free (B of step k-2 done)", "A: done, a_done raised"]):
    row(name, a[:, :, k] - t0)
for k, name in enumerate(["scan: reached the step (prev counts read)", "scan: phase A is through", "scan: lower counts known", "scan: games dealt, flag raised"]):
    row(name, scan[:, k] - t0)
for k, name in enumerate(["B: reached the step", "B: phase A is through", "B: movers' rows issued", "B: done (new rows too), b_done raised"]):
    row(name, bw[:, :, k] - t0)
print("durations, medians: A compute %.2f, scan look-back %.2f, scan deal %.2f, B movers' rows %.2f, B whole %.2f" % (
    np.median(a[:, :, 3] - a[:, :, 2]), np.median(scan[:, 2] - scan[:, 1]), np.median(scan[:, 3] - scan[:, 2]),
    np.median(bw[:, :, 2] - bw[:, :, 1]), np.median(bw[:, :, 3] - bw[:, :, 1])))
