"""Per-wave phase timeline of the TWO-LAUNCH Hanabi step, mrl_hanabi_step + mrl_hanabi_reset (what the sharded path runs;
tools/stamps_hanabi_fused.py is the single launch).  Diagnostic build: make -C madrona_rl_envs_playground_amd/csrc diag."""
import os, sys, torch, numpy as np
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import _lib
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
with _lib.debug_knobs({"stamps": 1, "fused_step": 2}):
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                          max_information_tokens=8, max_life_tokens=3)
assert sim.kernel_name == "mrl_hanabi_step", sim.kernel_name
mask, act = sim.action_mask_tensor().to_torch(), sim.action_tensor().to_torch()
for i in range(150):
    act.copy_((torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True).to(torch.int32))
    sim.step()
torch.cuda.synchronize()
st = sim._tensor(14).to_torch().cpu().numpy().view(np.uint64).reshape(-1, 16).astype(np.int64)
st = st[st[:, 0] > 0]
names = ["start", "loaded", "applied", "encoded", "phaseA done", "expanded", "records stored"]
print("s_memtime deltas per wave (cycles; /2400 = us), median [p10..p90] over", len(st), "waves (first sub-block)")
for a in range(6):
    d = st[:, a + 1] - st[:, a]
    print(f"{names[a]:>14s} -> {names[a+1]:14s} {np.median(d):8.0f}  [{np.percentile(d,10):7.0f} .. {np.percentile(d,90):7.0f}]  {np.median(d)/2400:5.2f} us")
full = st[(st[:, 7] > 0) & (st[:, 11] > 0)]
rnames = {7: "reset start", 8: "compacted", 9: "dealt", 10: "encoded", 11: "rows written"}
print(f"re-deal launch ({len(full)} waves of workgroups with finished worlds; 'compacted' includes the prefix sum)")
for a, b in zip([7, 8, 9, 10], [8, 9, 10, 11]):
    d = full[:, b] - full[:, a]
    print(f"{rnames[a]:>14s} -> {rnames[b]:14s} {np.median(d):8.0f}  [{np.percentile(d,10):7.0f} .. {np.percentile(d,90):7.0f}]  {np.median(d)/2400:5.2f} us")
if len(full):
    rs15, r12 = full[:, 15] / 100.0, full[:, 12] / 100.0
    t15 = st[st[:, 15] > 0][:, 15].min() / 100.0
    print(f"re-deal wave starts (us after the first): p50 {np.median(rs15 - t15):.2f} max {(rs15 - t15).max():.2f}")
    print(f"re-deal wave ends (us after the first start): p50 {np.median(r12 - t15):.2f} max {(r12 - t15).max():.2f}")
    print(f"step launch end -> re-deal launch start: {t15 - st[:, 14].max() / 100.0:.2f} us")
d = st[:, 6] - st[:, 0]
print(f"sub-block median {np.median(d):.0f} cycles = {np.median(d)/2400:.2f} us, max {d.max()/2400:.2f} us")
rs, re = st[:, 13] / 100.0, st[:, 14] / 100.0   # s_memrealtime: 100 MHz, common to the whole chip
t0 = rs.min()
print(f"wave starts (us after the first): p50 {np.median(rs - t0):.2f} p90 {np.percentile(rs - t0, 90):.2f} max {(rs - t0).max():.2f}")
print(f"wave ends   (us after the first start): p10 {np.percentile(re - t0, 10):.2f} p50 {np.median(re - t0):.2f} max {(re - t0).max():.2f}")
