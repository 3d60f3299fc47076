#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the Overcooked step, random policy.

Contract (one JSON line on rank 0):
    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: Overcooked cramped_room, 32768 worlds per
GPU, horizon 400, uniform random actions (scripts/overcooked_example.py:99-116
of the reference: `randint(high=6)` per agent per step, sampling outside the
timed call).  A step = one `mrl_step_with_actions` launch over the rank's
world shard, actions already resident in HBM (a pre-sampled pool, cycled).
Worlds are independent, so ranks do not communicate inside the timed region
(weak scaling, 32768 worlds per GPU); `--gather-obs` adds the RCCL all-gather of
the observation shards for consumers that need the global batch on every rank.

Also reported: `roofline` (achieved algorithmic HBM GB/s of the step kernel,
timed per launch with HIP events on the launch stream) and `cpu_baseline` (the
test-only CPU oracle in oracle/, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--worlds", type=int, default=32768, help="worlds per GPU")
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--horizon", type=int, default=400)
    ap.add_argument("--gather-obs", action="store_true", help="all-gather observation shards every step (RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused-steps", type=int, default=1000, help="steps of the extra device-side random rollout (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--pool", type=int, default=64, help="pre-sampled action tensors cycled through")
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota
    (a GPU box hands one GPU's job a share of the host, not all of it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(params, seconds):
    """TEST-ONLY oracle timed as the CPU baseline (kind 'port'): same layout,
    same action distribution, the host cores available to this job, a bounded
    number of world-steps."""
    from oracle import oracle
    cores = min(host_cores(), 64)
    n = 32768
    orc = oracle.OvercookedOracle(params, n, num_threads=cores)
    rng = np.random.default_rng(0)
    acts = [rng.integers(0, 6, size=(params["num_players"], n)).astype(np.int32) for _ in range(8)]
    for i in range(2):
        orc.step(acts[i])
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.step(acts[steps % len(acts)])
        steps += 1
    dt = time.perf_counter() - t0
    orc.close()
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps x {n} worlds of the same workload ({dt:.1f} s, OpenMP over worlds)"}


def main():
    args = parse()
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus and world_size > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
    if args.gpus > 1 and world_size == 1:
        raise SystemExit("launch multi-GPU runs with python -m torch.distributed.run --nproc-per-node N")

    import torch.distributed as dist
    from madrona_rl_envs_playground_amd import layouts
    from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator

    # Rehearsal on a one-GPU box (never the measured configuration): MRL_BENCH_REHEARSE=1 puts every rank
    # on device 0 and runs the rank protocol over gloo, so the multi-rank control flow can be exercised
    # where only one card exists.  The driver's real runs use one rank per GPU over RCCL.
    rehearse = world_size > 1 and os.environ.get("MRL_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    params = layouts.get_base_layout_params(args.layout, args.horizon)
    P, n = params["num_players"], args.worlds
    sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n, **params)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(args.pool)]
    obs = sim.observation_world_major_tensor().to_torch()
    gathered = None
    if args.gather_obs and world_size > 1 and not rehearse:
        gathered = torch.empty((world_size,) + tuple(obs.shape), dtype=obs.dtype, device=obs.device)

    def one_step(i):
        sim.step_with_actions(pool[i % args.pool])
        if gathered is not None:
            dist.all_gather_into_tensor(gathered, obs)

    def fence():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    fence()
    dt = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # average launch duration of the step kernel: HIP events on the launch stream (torch's
    # current stream is the stream the C ABI is handed) around K back-to-back launches.  The
    # queue never drains, so elapsed / K is the kernel's duration (rocprofv3 --kernel-trace
    # agrees within 1%, profiles/); bracketing every launch with its own event pair would add
    # ~2 us of marker overhead to a ~10 us kernel.
    k_launch = args.steps
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(k_launch):
        sim.step_with_actions(pool[i % args.pool])
    ev1.record()
    torch.cuda.synchronize()
    kernel_ms_avg = ev0.elapsed_time(ev1) / k_launch

    # extra, never `value`: the same random-policy workload with the actions drawn in the kernel and
    # the worlds' state kept in LDS between steps (mrl_rollout_random)
    fused = None
    if world_size == 1 and args.fused_steps > 0:
        sim.rollout_random(args.fused_steps, seed=99, first_step=0)
        torch.cuda.synchronize()
        ev0.record()
        sim.rollout_random(args.fused_steps, seed=99, first_step=args.fused_steps)
        ev1.record()
        torch.cuda.synchronize()
        fused_ms = ev0.elapsed_time(ev1)
        fused = {"value": n * args.fused_steps / (fused_ms * 1e-3), "unit": "env-steps/s", "steps_per_launch": args.fused_steps,
                 "us_per_step": fused_ms * 1e3 / args.fused_steps,
                 "note": "every step still writes its observation slab, rewards and dones"}

    # extra, never `value`: the SAME workload (the pre-sampled resident action pool, in order) as one open-loop
    # sequence per launch (mrl_step_sequence): what the per-step launches and the state round trips cost
    seq = None
    if world_size == 1 and args.fused_steps > 0:
        k_seq = min(args.fused_steps, 512)
        seq_actions = torch.stack([pool[i % args.pool] for i in range(k_seq)]).contiguous()
        sim.step_sequence(seq_actions)
        torch.cuda.synchronize()
        ev0.record()
        sim.step_sequence(seq_actions)
        ev1.record()
        torch.cuda.synchronize()
        seq_ms = ev0.elapsed_time(ev1)
        seq = {"value": n * k_seq / (seq_ms * 1e-3), "unit": "env-steps/s", "steps_per_launch": k_seq, "us_per_step": seq_ms * 1e3 / k_seq,
               "note": "same resident action pool, one launch per sequence; every step still writes observations, rewards, dones"}
        del seq_actions

    if rank == 0:
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run
        # inside this process); only reported when they were taken on this exact workload
        traffic = None
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "overcooked_step_traffic.json")))
            if pmc["layout"] == args.layout and pmc["worlds"] == n and pmc["kernel"] == sim.kernel_name:
                traffic = pmc["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        bytes_per_launch = sim.bytes_per_world_step * n
        achieved = bytes_per_launch / (kernel_ms_avg * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (whole node), Overcooked 32768 worlds, random policy",
            "value": n * world_size * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"Overcooked {args.layout}, {n} worlds per GPU, horizon {args.horizon}, "
                                   f"uniform random actions (pre-sampled pool of {args.pool}, resident in HBM)",
                       "worlds_per_gpu": n, "obs_gather": bool(gathered is not None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": sim.kernel_name, "kernel_us_avg": kernel_ms_avg * 1e3,
                         "bytes_per_launch": bytes_per_launch, "launches_timed": k_launch},
        }
        if fused is not None:
            out["fused_random_rollout"] = fused
        if seq is not None:
            out["action_sequence_per_launch"] = seq
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(params, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    sim.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
