"""GPU: the `nccl` (= RCCL) code path, executed at world_size 1 on the one GPU a test box has.

Every multi-GPU piece of this repo -- bench.py's rank protocol and obs_gather leg, `gather_worlds`,
`ShardedSimulator`'s episode exchange -- is written against torch.distributed with the `nccl` backend; a node with
several GPUs is the driver's, not the builder's.  A process group of ONE rank still goes through
`init_process_group("nccl", device_id=...)`, communicator creation, `barrier`, `all_reduce` / `all_gather` on device
tensors and `all_gather_into_tensor` straight into the global observation tensor, so a typo in any of them fails here
and not in the driver's 8-GPU run.  (The multi-rank control flow is covered by the gloo rehearsals in
test_gpu_configs.py and test_distributed_gloo.py.)  RCCL refuses two ranks on one device, so world_size 1 it is.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_rank_protocol_over_nccl_at_world_size_one(hip_lib):
    """`MRL_BENCH_FORCE_DIST=1 python bench.py --gpus 1`: one rank under torch.distributed.run, backend nccl; the
    line carries `ranks.backend == "nccl"` and the obs_gather leg (all_gather_into_tensor on device memory)."""
    env = dict(os.environ, MRL_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MRL_BENCH_REHEARSE"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "5"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), proc.stdout  # stdout is the JSON line and nothing else (RCCL's banner goes to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 50 and out["scaling"] == "weak"
    assert out["ranks"] == {"world_size": 1, "backend": "nccl", "rehearsal_on_one_gpu": False}
    g = out["obs_gather"]
    assert "RCCL" in g["collective"] and g["value"] > 0 and g["steps"] == 50
    assert g["bytes_per_rank_per_step"] == 32768 * 1040 and len(g["per_rank_ms_per_step"]) == 1
    assert out["value"] > 0 and out["roofline"]["bound"] == "hbm" and out["roofline"]["launches_timed"] >= 300
    assert len(out["timing"]["per_rank_ms_per_step"]) == 1


def _sharded_rank(rank, port, out_path):
    """The one rank of a world_size-1 `nccl` group: sharded simulators (phase 1 -> all-gather of SHARD_COUNT over RCCL
    -> mrl_step_phase2_gathered) next to plain ones fed the same actions; every exported tensor must stay equal."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from madrona_rl_envs_playground_amd import layouts
    from madrona_rl_envs_playground_amd.distributed import ShardedSimulator
    from madrona_rl_envs_playground_amd.simulators import (BalanceBeamSimulator, CartpoleSimulator, ExecMode, HanabiSimulator,
                                                           OvercookedSimulator)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=1, device_id=torch.device("cuda", 0))
    report = {"backend": dist.get_backend()}
    try:
        gen = torch.Generator(device="cuda").manual_seed(5)

        # ---- Hanabi: masked-random policy (scripts/hanabi_example.py:64-67), several episodes end inside the run ----
        n = 6000
        cfg = dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3)  # envs/hanabi_env.py:16-28
        make = lambda k: HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **cfg)  # noqa: E731
        sh, plain = ShardedSimulator(make, n), make(n)
        assert (sh.lo, sh.n, sh.world_size) == (0, n, 1)
        names = ["observation_tensor", "agent_state_tensor", "action_mask_tensor", "reward_tensor", "done_tensor",
                 "active_agent_tensor", "game_tensor", "reset_count_tensor"]
        finished = 0
        for t in range(90):
            mask = plain.action_mask_tensor().to_torch()
            a = (torch.rand(mask.shape, device="cuda", generator=gen) * mask).argmax(-1).to(torch.int32).unsqueeze(-1).contiguous()
            plain.step_with_actions(a)
            sh.step(a)
            for name in names:
                assert torch.equal(getattr(plain, name)().to_torch(), getattr(sh.sim, name)().to_torch()), f"hanabi {name} step {t}"
            assert torch.equal(sh.sim.shard_count_tensor().to_torch(), plain.reset_count_tensor().to_torch())  # phase 1's count
            finished += int(plain.reset_count_tensor().to_torch().item())
        obs = sh.sim.observation_tensor().to_torch()
        full = sh.gather(obs, world_dim=1)  # strided view -> contiguous -> all_gather_into_tensor over RCCL
        assert full.shape == obs.shape and torch.equal(full, obs)
        report["hanabi_finished"] = finished
        sh.close()
        plain.close()

        # ---- Cartpole ----
        n = 200_000
        sh, plain = ShardedSimulator(lambda k: CartpoleSimulator(ExecMode.CUDA, 0, k), n), CartpoleSimulator(ExecMode.CUDA, 0, n)
        finished = 0
        for t in range(120):
            a = torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda", generator=gen)
            plain.step_with_actions(a)
            sh.step(a)
            for name in ("observation_tensor", "reset_tensor", "reward_tensor", "reset_count_tensor"):
                assert torch.equal(getattr(plain, name)().to_torch(), getattr(sh.sim, name)().to_torch()), f"cartpole {name} step {t}"
            finished += int(plain.reset_count_tensor().to_torch().item())
        state = sh.sim.observation_tensor().to_torch()
        out = torch.empty_like(state)
        assert sh.gather(state, world_dim=0, out=out).data_ptr() == out.data_ptr() and torch.equal(out, state)
        report["cartpole_finished"] = finished
        sh.close()
        plain.close()

        # ---- balance beam (a third of the worlds re-seed every step) ----
        n = 50_000
        sh, plain = ShardedSimulator(lambda k: BalanceBeamSimulator(ExecMode.CUDA, 0, k), n), BalanceBeamSimulator(ExecMode.CUDA, 0, n)
        for t in range(30):
            a = torch.randint(0, 4, (2, n, 1), dtype=torch.int32, device="cuda", generator=gen)
            plain.step_with_actions(a)
            sh.step(a)
            for name in ("observation_tensor", "done_tensor", "reward_tensor", "reset_count_tensor"):
                assert torch.equal(getattr(plain, name)().to_torch(), getattr(sh.sim, name)().to_torch()), f"balance {name} step {t}"
        sh.close()
        plain.close()

        # ---- Overcooked: no episode exchange; the world-major observation slab gathered into a caller tensor ----
        params = layouts.get_base_layout_params("cramped_room", 50)
        n = 4096
        sh = ShardedSimulator(lambda k: OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **params), n,
                              needs_episode_exchange=False)
        slab = sh.sim.observation_world_major_tensor().to_torch()
        gathered = torch.empty_like(slab)
        for t in range(60):
            sh.step(torch.randint(0, 6, (2, n, 1), dtype=torch.int32, device="cuda", generator=gen))
            sh.gather(slab, world_dim=0, out=gathered)
            assert torch.equal(gathered, slab), f"overcooked gather step {t}"
        rew = sh.gather(sh.sim.reward_tensor().to_torch(), world_dim=1)
        assert torch.equal(rew, sh.sim.reward_tensor().to_torch())
        sh.close()
        torch.cuda.synchronize()
        report["ok"] = True
    finally:
        dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(report, f)


def test_sharded_equals_unsharded_under_nccl(hip_lib, tmp_path):
    """ShardedSimulator's exchange over RCCL on device memory: Hanabi, Cartpole and the balance beam hold exactly what
    one simulator of the whole batch holds; the Overcooked slab is all-gathered into the caller's tensor."""
    out = os.path.join(str(tmp_path), "report.json")
    mp.start_processes(_sharded_rank, args=(_free_port(), out), nprocs=1, join=True, start_method="spawn")
    report = json.load(open(out))
    assert report["ok"] and report["backend"] == "nccl"
    assert report["hanabi_finished"] > 50 and report["cartpole_finished"] > 1000  # the exchange had something to number
