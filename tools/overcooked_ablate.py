"""Diagnostic build only: Overcooked step time with phases ablated (MRL_ABLATE bits: 4 = no transition,
2 = encode without the HBM stores, 8 = no encode).  Timing only -- ablated runs compute garbage."""
import os, sys, torch
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import layouts
from madrona_rl_envs_playground_amd import _lib
_lib.debug_set("ablate", int(os.environ.get("MRL_ABLATE", "0")))  # the tool's own command-line knob
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
params = layouts.get_base_layout_params("cramped_room", 400)
sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
pool = [torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
for i in range(50):
    sim.step_with_actions(pool[i % 8])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(1000):
    sim.step_with_actions(pool[i % 8])
e1.record()
torch.cuda.synchronize()
print("MRL_ABLATE", os.environ.get("MRL_ABLATE", "0"), f"{e0.elapsed_time(e1):.2f} us per step at {n} worlds")
