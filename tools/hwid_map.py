"""Diagnostic build only: where the waves of the Overcooked step run (HW_ID / XCC_ID read in the kernel).
Prints how a workgroup's four waves spread over the SIMDs of its CU and which workgroups share a CU."""
import os, sys, collections, torch, numpy as np
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MRL_ENVS_LIB", os.path.join(_REPO, "madrona_rl_envs_playground_amd", "diag", "libmrl_envs_diag.so"))
sys.path.insert(0, _REPO)
from madrona_rl_envs_playground_amd import layouts, _lib
_lib.debug_set("stamps", 1)
from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator
n = 32768
params = layouts.get_base_layout_params("cramped_room", 400)
sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=n, **params)
pool = [torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda") for _ in range(8)]
for rep in range(3):
    for i in range(20):
        sim.step_with_actions(pool[i % 8])
    torch.cuda.synchronize()
    st = sim._tensor(14).to_torch().cpu().numpy().view(np.uint64).reshape(-1, 16)
    hw = st[:, 12]
    hwid, xcc = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64), ((hw >> np.uint64(32)).astype(np.int64)) & 0xF
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 3
    nwg = len(hw) // 4
    pat = collections.Counter(tuple(int(v) for v in simd[4 * b:4 * b + 4]) for b in range(nwg))
    print("SIMDs of a workgroup's waves 0..3:", pat.most_common(6))
    cukey = xcc * 1000 + se * 100 + sh * 50 + cu
    per_cu = collections.defaultdict(list)
    for b in range(nwg):
        per_cu[int(cukey[4 * b])].append(b)
    print("CUs used:", len(per_cu), " workgroups per CU:", dict(collections.Counter(len(v) for v in per_cu.values())))
    for k in list(per_cu)[:6]:
        bs = per_cu[k]
        print("  CU", k, "blockIdx:", bs, " wave0 SIMD:", [int(simd[4 * b]) for b in bs])
    for name, rule in (("b>>3", lambda b: (b >> 3) & 3), ("b>>8", lambda b: (b >> 8) & 3), ("b>>5", lambda b: (b >> 5) & 3),
                       ("b>>6", lambda b: (b >> 6) & 3), ("b>>7", lambda b: (b >> 7) & 3), ("0", lambda b: 0)):
        worst = collections.Counter()
        for k, bs in per_cu.items():
            c = collections.Counter(int(simd[4 * b + rule(b)]) for b in bs)
            worst[max(c.values())] += 1
        print(f"  leader = wave ({name})&3: most leaders on one SIMD of a CU -> {dict(worst)}")
