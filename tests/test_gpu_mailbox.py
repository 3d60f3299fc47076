"""GPU: the collective-free exchange of a sharded batch's finished counts (mrl_exchange_* / mrl_step_exchanged).

Every rank owns a mailbox in device memory which its peers map through IPC handles; a step's count launch stores the
rank's word into every mailbox and phase 2 polls its own.  One GPU is what a test box has: the protocol runs with ONE
rank (its own mailbox only) and with TWO PROCESSES that share the card -- the handles cross the process boundary through
hipIpcGetMemHandle / hipIpcOpenMemHandle exactly as they would between GPUs; what this cannot show is the visibility of a
store that arrives over xGMI (DESIGN.md section 6 says so).  Reference: one process-wide atomic hands out episode numbers,
src/hanabi_env/sim.cpp:449-451, src/cartpole_env/sim.cpp:51-53."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

HANABI = dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_one_rank_mailbox_equals_the_plain_step(hip_lib):
    """world_size 1: a shard that is the whole batch, stepped through its mailbox, holds what a plain simulator holds."""
    from madrona_rl_envs_playground_amd.distributed import ShardedSimulator
    from madrona_rl_envs_playground_amd.simulators import BalanceBeamSimulator, CartpoleSimulator, ExecMode, HanabiSimulator
    gen = torch.Generator(device="cuda").manual_seed(3)
    n = 5000
    make = lambda k: HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **HANABI)  # noqa: E731
    sh, plain = ShardedSimulator(make, n, exchange="mailbox"), make(n)
    names = ["observation_tensor", "agent_state_tensor", "action_mask_tensor", "reward_tensor", "done_tensor", "active_agent_tensor",
             "game_tensor", "reset_count_tensor"]
    finished = 0
    for t in range(80):
        mask = plain.action_mask_tensor().to_torch()
        a = (torch.rand(mask.shape, device="cuda", generator=gen) * mask).argmax(-1).to(torch.int32).unsqueeze(-1).contiguous()
        plain.step_with_actions(a)
        sh.step(a)
        for name in names:
            assert torch.equal(getattr(plain, name)().to_torch(), getattr(sh.sim, name)().to_torch()), f"hanabi {name} step {t}"
        finished += int(plain.reset_count_tensor().to_torch().item())
    assert finished > 50 and int(sh.sim.scan_timeout_tensor().to_torch().item()) == 0
    sh.close()
    plain.close()
    for make, high, shape, names in (
            (lambda k: CartpoleSimulator(ExecMode.CUDA, 0, k), 2, lambda k: (k, 1), ("observation_tensor", "reset_tensor", "reset_count_tensor")),
            (lambda k: BalanceBeamSimulator(ExecMode.CUDA, 0, k), 4, lambda k: (2, k, 1), ("observation_tensor", "done_tensor", "reset_count_tensor"))):
        n = 70001
        sh, plain = ShardedSimulator(make, n, exchange="mailbox"), make(n)
        for t in range(60):
            a = torch.randint(0, high, shape(n), dtype=torch.int32, device="cuda", generator=gen)
            plain.step_with_actions(a)
            sh.step(a)
            for name in names:
                assert torch.equal(getattr(plain, name)().to_torch(), getattr(sh.sim, name)().to_torch()), f"{name} step {t}"
        sh.close()
        plain.close()


def test_exchange_calls_are_validated(hip_lib):
    from madrona_rl_envs_playground_amd import _lib
    from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode
    sim = CartpoleSimulator(ExecMode.CUDA, 0, 100)
    with pytest.raises(_lib.MrlError, match="mailbox"):
        sim.step_exchanged()
    with pytest.raises(_lib.MrlError, match="rank"):
        sim.exchange_create(17, 0)
    with pytest.raises(_lib.MrlError, match="rank"):
        sim.exchange_create(2, 2)
    handle = sim.exchange_create(1, 0)
    assert len(handle) == _lib.IPC_HANDLE_BYTES
    with pytest.raises(_lib.MrlError, match="already"):
        sim.exchange_create(1, 0)
    sim.exchange_connect([handle])
    sim.step_exchanged()
    sim.close()


def _mailbox_rank(rank, ws, port, game, total, steps, out_dir):
    """One of two processes sharing GPU 0: its shard of a `total`-world batch, the finished counts exchanged through the
    IPC-mapped mailboxes (gloo carries the 64-byte handles once and the final comparison data)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from madrona_rl_envs_playground_amd.distributed import ShardedSimulator
    from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, HanabiSimulator
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        if game == "hanabi":
            make = lambda k: HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=k, **HANABI)  # noqa: E731
        else:
            make = lambda k: CartpoleSimulator(ExecMode.CUDA, 0, k)  # noqa: E731
        sh = ShardedSimulator(make, total, exchange="mailbox")
        gen = torch.Generator().manual_seed(99)  # the same global random stream on both ranks
        snaps = []
        for t in range(steps):
            if game == "hanabi":
                # masked-random legal moves: the mask of the whole batch is needed to draw from the global stream
                mask = sh.gather(sh.sim.action_mask_tensor().to_torch().contiguous(), world_dim=1).cpu()
                a = (torch.rand(mask.shape, generator=gen) * mask).argmax(-1).to(torch.int32).unsqueeze(-1)
                sh.step(a[:, sh.lo:sh.lo + sh.n].contiguous().cuda())
                local = sh.sim.game_tensor().to_torch()
                dim = 0
            else:
                a = torch.randint(0, 2, (total, 1), dtype=torch.int32, generator=gen)
                sh.step(a[sh.lo:sh.lo + sh.n].contiguous().cuda())
                local = sh.sim.observation_tensor().to_torch()
                dim = 0
            if t % 7 == 0 or t == steps - 1:
                snaps.append(sh.gather(local, world_dim=dim).cpu())
        assert int(sh.sim.scan_timeout_tensor().to_torch().item()) == 0
        if rank == 0:
            torch.save(snaps, os.path.join(out_dir, "snaps.pt"))
        torch.cuda.synchronize()
        dist.barrier()  # nobody unmaps a mailbox a peer's kernel may still write
        sh.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("game,total,steps", [("cartpole", 200001, 60), ("hanabi", 3000, 50)])
def test_two_processes_exchange_through_ipc_mailboxes(game, total, steps, hip_lib, tmp_path):
    """Two ranks, two processes, one GPU: every world of the sharded run holds what it holds in ONE simulator of the whole
    batch fed the same actions -- so every finished world took the episode number the global order gives it."""
    from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, HanabiSimulator
    mp.start_processes(_mailbox_rank, args=(2, _free_port(), game, total, steps, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    snaps = torch.load(os.path.join(str(tmp_path), "snaps.pt"))
    gen = torch.Generator().manual_seed(99)
    whole = (HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=0, num_worlds=total, **HANABI) if game == "hanabi"
             else CartpoleSimulator(ExecMode.CUDA, 0, total))
    k, finished = 0, 0
    for t in range(steps):
        if game == "hanabi":
            mask = whole.action_mask_tensor().to_torch().cpu()
            a = (torch.rand(mask.shape, generator=gen) * mask).argmax(-1).to(torch.int32).unsqueeze(-1)
            whole.step_with_actions(a.cuda())
            now = whole.game_tensor().to_torch()
        else:
            whole.step_with_actions(torch.randint(0, 2, (total, 1), dtype=torch.int32, generator=gen).cuda())
            now = whole.observation_tensor().to_torch()
        finished += int(whole.reset_count_tensor().to_torch().item())
        if t % 7 == 0 or t == steps - 1:
            assert torch.equal(snaps[k], now.cpu()), f"{game}: the sharded batch differs from one simulator at step {t}"
            k += 1
    assert k == len(snaps) and finished > 100
    whole.close()
