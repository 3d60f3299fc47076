#!/usr/bin/env python3
"""Per-wave timeline of the single-launch Cartpole step (diagnostic build: make -C madrona_rl_envs_playground_amd/csrc diag).
Stamps are s_memrealtime (100 MHz, common to the chip): microseconds after the launch's first wave started."""
import os, sys, torch, numpy as np
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import _lib
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
knobs = {"stamps": 1, "fused_step": 1}
knobs.update({k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])})
with _lib.debug_knobs(knobs):
    sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n)
assert sim.kernel_name == "mrl_cartpole_step_fused", sim.kernel_name
pool = [torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
for i in range(100):
    sim.step_with_actions(pool[i % 8])
torch.cuda.synchronize()
st = sim._tensor(14).to_torch().cpu().numpy().view(np.uint64).reshape(-1, 4, 8).astype(np.float64) / 100.0  # (block, wave, stamp) in us
t0 = st[:, :, 0].min()
names = ["start", "state in, pose done", "count published, look-back asked", "expensive half done", "past the barrier (count acknowledged)",
         "look-back done (wave 0)", "stores issued", "re-seeded (end)"]
print(f"{st.shape[0]} workgroups x 4 waves, {n} worlds; us after the first wave's start: p10 / p50 / p90 / max")
for k, name in enumerate(names):
    v = st[:, :, k] - t0
    if k == 5:
        v = v[:, 0]
    if k == 7:
        v = v[v > 0]
    print(f"{name:>48s}  {np.percentile(v,10):6.2f} {np.median(v):6.2f} {np.percentile(v,90):6.2f} {v.max():6.2f}")
end = np.maximum(st[:, :, 6], st[:, :, 7]) - t0
print("end of workgroup by index decile:", np.round([end.max(1)[i * len(end) // 10:(i + 1) * len(end) // 10].mean() for i in range(10)], 2).tolist())
start = st[:, :, 0].min(1) - t0
print("start of workgroup by index decile:", np.round([start[i * len(start) // 10:(i + 1) * len(start) // 10].mean() for i in range(10)], 2).tolist())
lb = st[:, 0, 5] - st[:, 0, 3]
print("look-back finish - expensive half done (wave 0) by index decile:", np.round([lb[i * len(lb) // 10:(i + 1) * len(lb) // 10].mean() for i in range(10)], 2).tolist())
rel = st - t0
for k in (1, 2, 3):
    print(f"stamp {k} by wave (p50 / p90 / max):", [f"{np.median(rel[:, w, k]):.2f}/{np.percentile(rel[:, w, k], 90):.2f}/{rel[:, w, k].max():.2f}" for w in range(4)])
    print(f"stamp {k} (wave 0) by index decile:", np.round([rel[i * len(rel) // 10:(i + 1) * len(rel) // 10, 0, k].mean() for i in range(10)], 2).tolist())
gap = rel[:, 0, 2] - rel[:, :, 1].max(1)
print("wave 0: published - last wave's pose done: p10/p50/p90/max", np.round([np.percentile(gap, 10), np.median(gap), np.percentile(gap, 90), gap.max()], 2).tolist())
slow = np.argsort(-rel[:, 0, 2])[:12]
print("the 12 latest publishers (block, xcd = block % 8, stamps 0..3 of wave 0, pose-done of the 4 waves):")
for b in slow:
    print(f"  {b:5d} {b % 8} {np.round(rel[b, 0, :4], 2).tolist()} {np.round(rel[b, :, 1], 2).tolist()}")
