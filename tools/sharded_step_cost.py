#!/usr/bin/env python3
"""What the sharded step's own machinery costs on ONE rank, without a collective: mrl_step (the library's choice), the
two-launch pair, and ShardedSimulator.step at world_size 1 -- phase 1 + the one-workgroup count launch + phase 2 (no process
group), with a one-rank nccl group (all-gather of one int32 over RCCL + mrl_step_phase2_gathered), and with the
collective-free mailbox exchange (mrl_step_exchanged).  us per step."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd import _lib  # noqa: E402
from madrona_rl_envs_playground_amd.distributed import ShardedSimulator  # noqa: E402
from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, HanabiSimulator  # noqa: E402


def us(fn, reps=300):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    with_group = "--nccl" in sys.argv
    if with_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # (RCCL prints its version banner on stdout when the first communicator comes up: stdout carries the JSON only)
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    out = {"process_group": "nccl, world_size 1" if with_group else "none"}
    games = (("hanabi", 65536, lambda k: HanabiSimulator(ExecMode.CUDA, 0, k, 5, 5, 2, 8, 3)),
             ("cartpole", 1 << 20, lambda k: CartpoleSimulator(ExecMode.CUDA, 0, k)))
    for name, n, make in games:
        row = {"worlds": n}
        sim = make(n)
        row["mrl_step_us"] = us(lambda i: sim.step())
        row["mrl_step_kernel"] = sim.kernel_name
        sim.close()
        with _lib.debug_knobs({"fused_step": 2}):
            sim = make(n)
        row["two_launch_step_us"] = us(lambda i: sim.step())
        sim.close()
        sh = ShardedSimulator(make, n)
        row["sharded_step_us"] = us(lambda i: sh.step())  # without a process group: the plain step; with one: + the collective
        sh.close()
        # the two-phase machinery alone, no collective: phase 1 + the one-workgroup count launch + gathered phase 2
        sim = make(n)
        sim.reseed_shard(0, n)
        counts = sim.shard_count_tensor().to_torch()

        def two_phase(i):
            sim.step_phase1(None)
            sim.step_phase2_gathered(counts, 0)
        row["phase1_count_phase2_gathered_us"] = us(two_phase)
        sim.close()
        # the collective-free exchange (mrl_step_exchanged): phase 1 + count-and-publish launch + phase 2 polling its mailbox
        sh = ShardedSimulator(make, n, exchange="mailbox")
        row["sharded_step_mailbox_us"] = us(lambda i: sh.step())
        sh.close()
        out[name] = row
    print(json.dumps(out))
    if with_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
