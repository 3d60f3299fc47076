import os, sys, torch, numpy as np
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# needs the diagnostic build: make -C madrona_rl_envs_playground_amd/csrc diag
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import layouts
from madrona_rl_envs_playground_amd import _lib
_lib.debug_set("stamps", 1)
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator
# usage: tools/stamps.py [layout [worlds [knob=value ...]]]   (two groups per wave: the phase stamps are the FIRST group's, "end" the wave's)
layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    _lib.debug_set(k, int(v))
params = layouts.get_base_layout_params(layout, 400)
sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
print(layout, n, sys.argv[3:], sim.kernel_name)
pool = [torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
for i in range(50):
    sim.step_with_actions(pool[i % 8])
torch.cuda.synchronize()
st = sim._tensor(14).to_torch().cpu().numpy().view(np.uint64).reshape(-1, 16).astype(np.int64)
st = st[st[:, 0] > 0]
names = ["start", "loaded+sync", "transition", "curmap", "dynamic list", "patch+stream", "loads issued+fills", "state in LDS", "pass3", "", "", "", "", "", "", "end"]
cols = [k for k in (0, 6, 7, 1, 2, 3, 4, 5, 8, 9, 10, 11, 15) if (st[:, k] > 0).all()]  # (slot 12 holds HW_ID / XCC_ID, not a time)
print("s_memtime deltas per wave (shader cycles; /2400 = us at 2.4 GHz), median [p10..p90] over", len(st), "waves")
for a, b in zip(cols, cols[1:]):
    d = st[:, b] - st[:, a]
    print(f"{names[a]:>14s} -> {names[b]:14s} {np.median(d):8.0f}  [{np.percentile(d,10):7.0f} .. {np.percentile(d,90):7.0f}]  {np.median(d)/2400:5.2f} us")
d = st[:, 15] - st[:, 0]
print(f"wave lifetime median {np.median(d):.0f} cycles = {np.median(d)/2400:.2f} us, max {d.max()/2400:.2f} us")

rs, re = st[:, 13] / 100.0, st[:, 14] / 100.0   # s_memrealtime: 100 MHz, common to the whole chip
t0 = rs.min()
print(f"wave starts (us after the first): p50 {np.median(rs - t0):.2f} p90 {np.percentile(rs - t0, 90):.2f} max {(rs - t0).max():.2f}")
print(f"wave ends   (us after the first start): p10 {np.percentile(re - t0, 10):.2f} p50 {np.median(re - t0):.2f} max {(re - t0).max():.2f}")
# by XCD (XCC_ID, upper half of slot 12): does every XCD get the same share of the fabric?
xcc = (st[:, 12] >> 32) & 0xF
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  XCC {x}: {m.sum():5d} waves, start p50 {np.median(rs[m] - t0):5.2f}, end p50 {np.median(re[m] - t0):5.2f} p90 {np.percentile(re[m] - t0, 90):5.2f} max {(re[m] - t0).max():5.2f}")
