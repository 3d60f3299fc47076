"""Per-wave timeline of the single-launch Hanabi step (diagnostic build: make -C madrona_rl_envs_playground_amd/csrc diag).
All stamps are s_memrealtime (100 MHz, common to the chip): microseconds after the launch's first wave started."""
import os, sys, torch, numpy as np
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import _lib
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
knobs = {"stamps": 1}
if len(sys.argv) > 2:
    knobs["hanabi.pairing"] = int(sys.argv[2])  # 4 / 1 / 0: phase-A pairing; +256: workgroups 2k and 2k + 1 swap their worlds
with _lib.debug_knobs(knobs):
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                          max_information_tokens=8, max_life_tokens=3)
assert sim.kernel_name == "mrl_hanabi_step_fused", sim.kernel_name
mask, act = sim.action_mask_tensor().to_torch(), sim.action_tensor().to_torch()
for i in range(150):
    act.copy_((torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True).to(torch.int32))
    sim.step()
torch.cuda.synchronize()
st = sim._tensor(14).to_torch().cpu().numpy().view(np.uint64).reshape(-1, 8, 16).astype(np.float64) / 100.0  # (block, wave, stamp) in us
t0 = st[:, :, 0].min()
names = {0: "start", 1: "records in LDS", 2: "phase A done", 3: "past the barrier", 5: "rows issued", 6: "records stored (end)"}
print(f"{st.shape[0]} workgroups x 8 stepping waves; us after the first wave's start: p10 / p50 / p90 / max")
def row(name, v):
    print(f"{name:>34s}  {np.percentile(v,10):6.2f} {np.median(v):6.2f} {np.percentile(v,90):6.2f} {v.max():6.2f}")
for k, name in names.items():
    row(name, st[:, :, k] - t0)
    if k == 1:
        row("leaders: transition done", st[:, :4, 4] - t0)
        row("leaders: encode done", st[:, :4, 2] - t0)
        row("leaders: line 0 handed over", st[:, :4, 11] - t0)
        row("partners: line 0 flag seen", st[:, 4:, 11] - t0)
        row("partners: first lines issued", st[:, 4:, 4] - t0)
        row("leaders: lines 3, 4 handed over", st[:, :4, 12] - t0)
        row("partners: lines 3, 4 flag seen", st[:, 4:, 12] - t0)
scan = st[:, 0, 8:11] - t0  # the scan wave's stamps
row("scan wave: counts handed over", scan[:, 0])
row("scan wave: prefix known, visible", scan[:, 1])
row("scan wave: re-deal done (end)", scan[:, 2])
end = st[:, :, 6] - t0
print(f"end of the last stepping wave per workgroup: p10 {np.percentile(end.max(1),10):.2f} p50 {np.median(end.max(1)):.2f} max {end.max():.2f}")
print(f"spread of wave ends inside a workgroup (max - min): p50 {np.median(end.max(1)-end.min(1)):.2f} max {(end.max(1)-end.min(1)).max():.2f}")
print(f"which wave ends last: {np.bincount(end.argmax(1), minlength=8).tolist()}")
by_xcd = end.max(1).reshape(-1)[: (st.shape[0] // 8) * 8].reshape(-1, 8)
print("mean workgroup end by blockIdx % 8 (XCD):", np.round(by_xcd.mean(0), 2).tolist())
