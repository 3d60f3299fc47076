"""many_player_layout (15x17) at 1000 worlds: step time by player count, launch shape, and (diagnostic build,
MRL_ABLATE bits: 1 = no row assembly, 2 = no HBM stores, 4 = no transition, 8 = no encode) where it goes."""
import os, sys, json, torch
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("MRL_ABLATE"):
    os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import layouts, _lib
if os.environ.get("MRL_ABLATE"):
    _lib.debug_set("ablate", int(os.environ["MRL_ABLATE"]))
for kv in os.environ.get("MRL_KNOBS", "").split(","):
    if kv:
        k, v = kv.split("=")
        _lib.debug_set(k, int(v))
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
out = []
for P in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["8", "16", "30"])]:
    params = layouts.get_base_layout_params("many_player_layout", 400, max_num_players=P)
    sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
    pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda") for _ in range(4)]
    for i in range(5):
        sim.step_with_actions(pool[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 100
    e0.record()
    for i in range(steps):
        sim.step_with_actions(pool[i % 4])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    obs_bytes = P * 255 * (5 * P + 16) * n
    out.append({"players": P, "us_per_step": us, "obs_GB": obs_bytes / 1e9, "obs_TBps": obs_bytes / us / 1e6,
                "launch_shape": list(sim.launch_shape), "kernel": sim.kernel_name})
    sim.close()
print(json.dumps({"ablate": os.environ.get("MRL_ABLATE", "0"), "knobs": os.environ.get("MRL_KNOBS", ""), "rows": out}))
