"""Diagnostic build only: time of the fused Hanabi step with phases ablated (MRL_ABLATE: 1 = no
action/encode, 2 = no row stores).  Worlds are first advanced 30 legal steps with the full kernel
semantics of the given build, so ablated runs measure timing only."""
import os, sys, torch
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import _lib
_lib.debug_set("ablate", int(os.environ.get("MRL_ABLATE", "0")))  # the tool's own command-line knob
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator
n = 65536
sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                      max_information_tokens=8, max_life_tokens=3)
for i in range(20):
    sim.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(200):
    sim.step()
e1.record()
torch.cuda.synchronize()
print("MRL_ABLATE", os.environ.get("MRL_ABLATE", "0"), f"{e0.elapsed_time(e1) * 5:.2f} us per step (back-to-back launches, action 0 everywhere)")
