"""Times the Hanabi step + reset kernels of one library build (MRL_ENVS_LIB) with HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd.simulators import ExecMode, HanabiSimulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, colors=5, ranks=5, players=2,
                      max_information_tokens=8, max_life_tokens=3)
mask, act = sim.action_mask_tensor().to_torch(), sim.action_tensor().to_torch()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t1 = t2 = 0.0
resets = 0
for i in range(140):
    act.copy_((torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True).to(torch.int32))
    e[0].record(); sim.step_phase1(); e[1].record(); sim.step_phase2(); e[2].record()
    torch.cuda.synchronize()
    if i >= 40:
        t1 += e[0].elapsed_time(e[1]); t2 += e[1].elapsed_time(e[2])
        resets += int(sim._tensor(10).to_torch().item())
print(os.environ.get("MRL_ENVS_LIB", "default").split("/")[-1], f"step {t1*10:.1f} us  reset {t2*10:.1f} us  (events around single launches: +~3 us each)  resets/step {resets/100:.0f}")
