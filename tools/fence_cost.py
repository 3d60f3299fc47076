#!/usr/bin/env python3
"""What the fence around a timed block costs on an idle GPU: torch.cuda.synchronize() alone, dist.barrier() + synchronize, and a
one-element all_reduce enqueued on the stream + synchronize (the same rendezvous without ProcessGroupNCCL::barrier's extra
host-side synchronisations).  One-rank nccl group on the one GPU the builder has; us per fence."""
import os, sys, time, json
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
token = torch.zeros(1, device="cuda")
x = torch.zeros(1 << 20, device="cuda")


def cost(fence, reps=300):
    for _ in range(20):
        fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        x.add_(1.0)  # something on the stream, as after a block of steps
        fence()
    return (time.perf_counter() - t0) / reps * 1e6


def f_sync():
    torch.cuda.synchronize()


def f_barrier():
    dist.barrier()
    torch.cuda.synchronize()


def f_allreduce():
    dist.all_reduce(token)
    torch.cuda.synchronize()


print(json.dumps({"synchronize_us": cost(f_sync), "barrier_plus_synchronize_us": cost(f_barrier), "all_reduce_plus_synchronize_us": cost(f_allreduce)}))
dist.destroy_process_group()
