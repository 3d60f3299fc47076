"""Times the Hanabi step of one library build (MRL_ENVS_LIB) under the device-side random policy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                      max_information_tokens=8, max_life_tokens=3)
sim.rollout_random(50, seed=3, first_step=0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
sim.rollout_random(300, seed=3, first_step=50)
e1.record()
torch.cuda.synchronize()
print(os.environ.get("MRL_ENVS_LIB", "default").split("/")[-1], f"{e0.elapsed_time(e1) / 300 * 1e3:.2f} us per step, timeout flag",
      int(sim.scan_timeout_tensor().to_torch().item()))
