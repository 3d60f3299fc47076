#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the Overcooked step, random policy.

Contract (one JSON line on rank 0):
    python bench.py --gpus N --steps K --warmup W
For N > 1 without a launcher this process only SPAWNS `python -m torch.distributed.run --nnodes=1
--nproc-per-node N ... bench.py <same flags>` (before anything touches the GPU) and returns its exit
code; under a launcher (WORLD_SIZE set) it is one rank of the job.

Workload = BASELINE.json configs[1]: Overcooked cramped_room, 32768 worlds per GPU, horizon 400,
uniform random actions (scripts/overcooked_example.py:99-116 of the reference: `randint(high=6)` per
agent per step, sampling outside the timed call).  A step = one `mrl_step_with_actions` launch over
the rank's world shard, actions already resident in HBM (a pre-sampled pool, cycled).  Worlds are
independent, so ranks do not communicate inside the timed region of `value` (weak scaling, 32768
worlds per GPU).  N > 1 also times the `obs_gather` leg of BASELINE.json configs[3]: the same step
followed by the RCCL all-gather of the world-major observation shards (34 MB per rank per step).

`MRL_BENCH_FORCE_DIST=1 python bench.py --gpus 1 ...` runs the N > 1 protocol at world_size 1 (it starts ONE
rank under torch.distributed.run): process group over `nccl` (= RCCL), the fences' barrier, the max-over-ranks
all_reduce / all_gather on device tensors and the obs_gather leg's all_gather_into_tensor all execute on the
one GPU a test box has (tests/test_gpu_nccl.py); the headline extras are skipped in that mode.

Timing: W warm-up steps, then blocks of exactly K steps, each bracketed by barrier +
`torch.cuda.synchronize()` on both sides, max over ranks.  One block is the contract; when K steps
are shorter than 50 ms (K = 20 is 0.2 ms here) the block is repeated and the MEDIAN block is
reported (`timing.blocks` says how many), so that one scheduling hiccup is not the result.
`timing.empty_block_ms` is what a block of no steps costs between the same two fences -- the bracket's
share of a short block (at --steps 20 it is 6 % of the block on one GPU, more with the ranks' rendezvous
in it); it is reported, never subtracted.

Also on the line: `roofline` (algorithmic HBM bytes of the step kernel / its average launch duration,
HIP events on the launch stream over >= 300 back-to-back launches; spec and on-box measured peaks),
`cpu_baseline` (the test-only CPU oracle in oracle/ on this host's cores, bounded sample), the
reference harness's own two timed regions, `wrapped_n_step` (scripts/overcooked_example.py:104-106)
and `isolated_with_copies` (scripts/overcooked_isolated_example.py:56-65), and `other_configs`: the other
BASELINE.json configurations one GPU can run -- Cartpole 1024 worlds (configs[0]) and 1 M worlds, Hanabi
65536 worlds (configs[2]: scripts/hanabi_example.py:62-80), the four other standard layouts at the 32768-world
shard of configs[3], and the two sibling worlds (Simplecooked `simple` 32768 worlds, balance beam 1 M worlds) -- each with
its kernel, average launch duration over >= 300 launches, algorithmic bytes per world-step and roofline fraction; the Hanabi
and the 1 M-world Cartpole legs carry a `cpu_baseline` of their own (the oracle, ~2 s); `mappo_rollout_loop_32768` is
configs[4] on one GPU (50 loop steps of tools/mappo_rollout_loop.py: the PyTorch policy's loop time and, beside it, what this
engine owns of a loop step: `env_step_plus_buffer_insert_us`).  `build_hash` = `mrl_build_hash()` of the running library.
With N > 1 the line also carries `expected` (and `obs_gather.expected`): what DESIGN.md section 6 predicted before any
multi-GPU run existed.

`roofline.traffic` (and `traffic` of the other legs) are HBM bytes per launch from rocprofv3 PMC passes, which
cannot run inside this process: they come from profiles/step_traffic.json and are quoted only when that file's
`csrc_sha16` equals the hash compiled into the running library (`mrl_build_hash()`, also on the line as `build_hash`), null otherwise.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MIN_BLOCK_SECONDS = 0.05
MIN_ROOFLINE_LAUNCHES = 300


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--worlds", type=int, default=32768, help="worlds per GPU")
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--horizon", type=int, default=400)
    ap.add_argument("--no-gather-leg", action="store_true", help="N > 1: skip the obs all-gather leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: only the headline, roofline and cpu_baseline")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the legs of the other BASELINE configurations")
    ap.add_argument("--fused-steps", type=int, default=1000, help="steps of the extra device-side random rollout (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--pool", type=int, default=64, help="pre-sampled action tensors cycled through")
    ap.add_argument("--large-worlds", type=int, default=1 << 20, help="worlds of the large-batch roofline point (0 = skip)")
    return ap.parse_args(argv)


def force_dist():
    return os.environ.get("MRL_BENCH_FORCE_DIST") == "1"


def spawn_ranks(args):
    """--gpus N > 1 (or MRL_BENCH_FORCE_DIST=1) and no launcher: start the N ranks as a child job.  Nothing in THIS process has
    touched the GPU (torch is not even imported), and the child is a child, never an exec."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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
    import numpy as np
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


def cpu_baseline_game(game, n, seconds=2.0):
    """The TEST-ONLY oracle of Hanabi / Cartpole timed like `cpu_baseline` (kind 'port'): the leg's world count, the leg's
    policy (masked-random legal moves / uniform random pushes), the host cores available to this job, ~`seconds` of work
    (the reference quotes its CPU figures beside the GPU's: src/hanabi_env/README.org:77-81, src/cartpole_env/README.org:80-85)."""
    import numpy as np
    from oracle import oracle
    cores = min(host_cores(), 64)
    rng = np.random.default_rng(0)
    if game == "hanabi":
        orc = oracle.HanabiOracle(dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3), n, num_threads=cores)

        def one():  # scripts/hanabi_example.py:64-67: argmax(rand * mask) of each agent's current mask
            orc.step((rng.random(orc.mask.shape, dtype=np.float32) * (orc.mask != 0)).argmax(-1).astype(np.int32))
    else:
        orc = oracle.CartpoleOracle(n, num_threads=cores)
        acts = [rng.integers(0, 2, size=(n, 1)).astype(np.int32) for _ in range(8)]
        count = [0]

        def one():
            orc.step(acts[count[0] % 8])
            count[0] += 1
    one()
    steps, in_oracle, t0 = 0, 0.0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        one()
        steps += 1
    dt = time.perf_counter() - t0
    orc.close()
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps x {n} worlds ({dt:.1f} s, OpenMP over worlds"
                      + (", the numpy sampling of the legal moves included)" if game == "hanabi" else ")")}


def mappo_leg(args, torch, local_rank):
    """configs[4] on one GPU: the loop of train/MAPPO/main_player.py:211-261 (tools/mappo_rollout_loop.py: fp32 torch CNN actor +
    critic for ego and partner, env.step, buffer insert) at the headline's batch.  The policy is PyTorch and out of this
    engine's scope; what this engine owns is `env_step_plus_buffer_insert_us`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mappo_rollout_loop", os.path.join(REPO, "tools", "mappo_rollout_loop.py"))
    loop = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loop)
    n, T, steps = args.worlds, 16, 50
    out = {"workload": f"MAPPO-style rollout loop, {args.layout}, {n} worlds", "steps": steps}
    for in_place in (True, False):
        env, ego, buffers = loop.build(args.layout, n, gpu_id=local_rank, horizon=args.horizon, steps_in_buffer=T, seed=0, in_place=in_place)
        ob = loop.rollout(env, ego, buffers, env.reset(), 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ob = loop.rollout(env, ego, buffers, ob, steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        key = "step_into_slot" if in_place else "step_then_clone_insert"
        out.setdefault("loop_ms_per_step", {})[key] = ms
        out.setdefault("loop_env_steps_per_s", {})[key] = n / (ms * 1e-3)
        if in_place:
            rand = torch.randint(0, 6, (env.num_players, n, 1), device="cuda")
            ring = buffers["obs"]
            copies = torch.empty((8, n, env.width, env.height, 5 * env.num_players + 16), dtype=torch.int8, device="cuda")

            def timed(fn, reps=60, blocks=5):
                # host-clocked and host-driven (one Python call per step): the median of a few short blocks, so that one
                # preemption of this process does not end up in the figure (a single 200-step block once read 66 us for 11)
                for i in range(10):
                    fn(i)
                took = []
                for _ in range(blocks):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for i in range(reps):
                        fn(i)
                    torch.cuda.synchronize()
                    took.append((time.perf_counter() - t) / reps * 1e6)
                return sorted(took)[len(took) // 2]
            out["env_step_plus_buffer_insert_us"] = {
                "step_into_slot": timed(lambda i: env.n_step(rand, out=ring[i % T])),
                "step_then_clone_insert": timed(lambda i: copies[i % 8].copy_(env.n_step(rand)[0][env.ego_ind].obs))}
            env.n_step(rand)
            del copies
        env.close()
        del buffers
    own = out["env_step_plus_buffer_insert_us"]["step_into_slot"]
    out["policy_share_of_loop"] = 1.0 - own * 1e-3 / out["loop_ms_per_step"]["step_into_slot"]
    out["note"] = ("loop_* is the PyTorch policy's number (fp32 CNN forward for ego and partner: ~99 % of a loop step); this engine's piece "
                   "is env_step_plus_buffer_insert_us -- the step writing its observations into the buffer slot against step + clone-insert")
    return out


def expected_gather(shard_bytes, world_size, step_seconds):
    """What the obs_gather leg should show on an 8-GPU MI355X node, written down BEFORE the first multi-GPU run (DESIGN.md
    section 6): every rank receives (G - 1) shards over its xGMI links (7 links per GPU, ~153 GB/s each by SURVEY.md section 5's
    figure, point to point).  Upper bound on the bus bandwidth: all G - 1 peers' links streaming at the link rate; lower: RCCL's
    ring (one link's worth, ~60 % efficient).  `gather_ms_per_step` follows from bytes / busbw."""
    peers = world_size - 1
    if peers == 0:
        return {"basis": "world_size 1: the collective is a device-to-device copy of the shard, no link is crossed"}
    recv = shard_bytes * peers
    hi = peers * 153.0          # direct all-gather, every peer on its own link at the link rate
    lo = 0.6 * 153.0            # ring: one link's rate at RCCL's usual efficiency
    ms = [recv / (bw * 1e9) * 1e3 for bw in (hi, lo)]
    return {"busbw_GBps": [lo, hi], "gather_ms_per_step": ms,
            "value": [world_size * 32768 / (step_seconds + m * 1e-3) for m in reversed(ms)],
            "basis": "bytes received per rank = shard x (G - 1); busbw between one xGMI link at 60 % (ring) and G - 1 links at 153 GB/s (direct)"}


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` on `workload` from the committed PMC passes -- only if they were taken on THIS build
    of the kernels (tools/pmc_traffic.py writes the file together with the hash of csrc/)."""
    from madrona_rl_envs_playground_amd import _lib
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "step_traffic.json")))
        if pmc["csrc_sha16"] != _lib.build_hash():
            return None
        for row in pmc["launches"]:
            if row["kernel"] == kernel and row["workload"] == workload:
                return row["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def other_configs(args, torch, local_rank, launches_us):
    """The other BASELINE.json configurations that fit one GPU, each as one launch (or the library's launches) per step
    with actions resident in HBM, timed like the headline's roofline: HIP events around back-to-back launches."""
    from madrona_rl_envs_playground_amd import _lib, layouts
    from madrona_rl_envs_playground_amd.simulators import CartpoleSimulator, ExecMode, HanabiSimulator, OvercookedSimulator
    launches = max(MIN_ROOFLINE_LAUNCHES, 300)
    out = {}

    def leg(sim, n, us, workload, note=None, steps_per_launch=1):
        gbps = sim.bytes_per_world_step * n / (us * 1e-6) / 1e9
        d = {"workload": workload, "worlds": n, "kernel": sim.kernel_name, "kernel_us_avg": us, "launches_timed": launches,
             "value": n / (us * 1e-6), "unit": "env-steps/s", "bytes_per_world_step": sim.bytes_per_world_step,
             "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS, "traffic": measured_traffic(sim.kernel_name, workload)}
        if note:
            d["note"] = note
        return d

    # configs[3]: the other four standard layouts at the per-GPU shard
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4321)
    for layout in ("asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit"):
        params = layouts.get_base_layout_params(layout, args.horizon)
        n, P = args.worlds, params["num_players"]
        sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n, **params)
        pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(8)]
        for i in range(20):
            sim.step_with_actions(pool[i % 8])
        us = launches_us(lambda i: sim.step_with_actions(pool[i % 8]), launches)
        out[f"overcooked_{layout}_{n}"] = leg(sim, n, us, f"overcooked {layout} {n}")
        sim.close()
        del pool

    # configs[0] (Cartpole 1024 worlds; the reference runs it on its CPU executor) and the same game at 1 M worlds
    for n in (1024, 1 << 20):
        sim = CartpoleSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n)
        pool = [torch.randint(0, 2, (n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(8)]
        for i in range(20):
            sim.step_with_actions(pool[i % 8])
        us = launches_us(lambda i: sim.step_with_actions(pool[i % 8]), launches)
        out[f"cartpole_{n}"] = leg(sim, n, us, f"cartpole {n}",
                                   note="kernel_us_avg = time per step call back to back; at 1024 worlds that is the host's call rate, not the kernel" if n == 1024 else None)
        sim.close()
        del pool
        if n > 1024 and not args.no_cpu_baseline:
            out[f"cartpole_{n}"]["cpu_baseline"] = cpu_baseline_game("cartpole", n)

    # the sibling worlds: Simplecooked (overcooked2_env, what the reference's trainers use) on its
    # `simple` layout at the 32768-world shard, the balance beam at 1 M worlds
    from madrona_rl_envs_playground_amd.simulators import BalanceBeamSimulator, SimplecookedSimulator
    params = layouts.get_simplecooked_layout_params("simple", args.horizon)
    n = args.worlds
    sim = SimplecookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n, **params)
    pool = [torch.randint(0, 6, (params["num_players"], n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(8)]
    for i in range(20):
        sim.step_with_actions(pool[i % 8])
    us = launches_us(lambda i: sim.step_with_actions(pool[i % 8]), launches)
    out[f"simplecooked_simple_{n}"] = leg(sim, n, us, f"simplecooked simple {n}")
    sim.close()
    n = 1 << 20
    sim = BalanceBeamSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n)
    pool = [torch.randint(0, 4, (2, n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(8)]
    for i in range(20):
        sim.step_with_actions(pool[i % 8])
    us = launches_us(lambda i: sim.step_with_actions(pool[i % 8]), launches)
    out[f"balance_beam_{n}"] = leg(sim, n, us, f"balance_beam {n}")
    sim.close()
    del pool

    # configs[2]: Hanabi full game, 65536 worlds.  The reference harness draws argmax(rand * mask) with torch ops between
    # the steps (scripts/hanabi_example.py:64-67): `harness_loop_us_per_step` is that loop; `kernel_us_avg` is the step
    # alone, one call per step, with the same uniformly-random-legal-move policy drawn by the step kernel itself
    # (mrl_rollout_random, one step per call; hanabi.no_persistent keeps it one ordinary launch per step).
    n = 65536
    with _lib.debug_knobs({"hanabi.no_persistent": 1}):
        sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n, colors=5, ranks=5, players=2,
                              max_information_tokens=8, max_life_tokens=3)
    for i in range(60):
        sim.rollout_random(1, seed=7, first_step=i)
    us = launches_us(lambda i: sim.rollout_random(1, seed=7, first_step=60 + i), launches)
    h = leg(sim, n, us, f"hanabi {n}", note="one step call per launch duration; uniformly random legal moves drawn by the step kernel")
    # bytes_per_world_step counts what this engine has to move: the OBSERVATION tensor is a view of the STATE row's first 658
    # bytes (the reference's state starts with a copy of its observation), so those bytes are written once.  The reference's
    # own format stores them twice: on THAT byte count the same launch reads as follows
    ref_bytes = sim.bytes_per_world_step + 658
    h["reference_format_bytes_per_world_step"] = ref_bytes
    h["frac_on_reference_format_bytes"] = ref_bytes * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS
    mask, act = sim.action_mask_tensor().to_torch(), sim.action_tensor().to_torch()

    def harness(i):
        act.copy_((torch.rand(mask.shape, device="cuda") * mask).argmax(-1, keepdim=True))
        sim.step()
    for i in range(10):
        harness(i)
    h["harness_loop_us_per_step"] = launches_us(harness, 200)
    sim.close()
    sim = HanabiSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n, colors=5, ranks=5, players=2,
                          max_information_tokens=8, max_life_tokens=3)
    sim.rollout_random(50, seed=7, first_step=0)
    h["persistent_rollout_us_per_step"] = launches_us(lambda i: sim.rollout_random(300, seed=7, first_step=50 + 300 * i), 2) / 300
    h["rollout_kernel"] = sim.rollout_kernel_name  # (the step kernel's name here = the runtime refused the cooperative launch)
    sim.close()
    if not args.no_cpu_baseline:
        h["cpu_baseline"] = cpu_baseline_game("hanabi", n)
    out[f"hanabi_{n}"] = h
    # configs[4]: the MAPPO rollout inner loop at the headline's batch
    out[f"mappo_rollout_loop_{args.worlds}"] = mappo_leg(args, torch, local_rank)
    return out


def run(args):
    import torch
    import torch.distributed as dist
    from madrona_rl_envs_playground_amd import _lib, layouts
    from madrona_rl_envs_playground_amd.distributed import gather_worlds
    from madrona_rl_envs_playground_amd.simulators import ExecMode, OvercookedSimulator

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")

    # Rehearsal on a one-GPU box (never the measured configuration): MRL_BENCH_REHEARSE=1 puts every rank
    # on device 0 and runs the rank protocol over gloo (observation gather staged through the host), so
    # the multi-rank control flow can be exercised where only one card exists.  The driver's real runs
    # use one rank per GPU over RCCL.
    rehearse = world_size > 1 and os.environ.get("MRL_BENCH_REHEARSE") == "1"
    # use_dist: the rank protocol (process group, barriers, reductions over ranks, gather leg) is in force.  Always for
    # N > 1; at N = 1 only under MRL_BENCH_FORCE_DIST=1, which is how the nccl path is executed on a one-GPU box.
    use_dist = world_size > 1 or force_dist()
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if rehearse else "nccl"
        # gloo's C++ side announces its connections and RCCL prints its version banner on stdout when the first communicator
        # comes up; stdout carries the one JSON line and nothing else, so both go to stderr
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()  # brings the communicator up now, inside the redirection
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    params = layouts.get_base_layout_params(args.layout, args.horizon)
    P, n = params["num_players"], args.worlds
    sim = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=n, **params)
    C, F = sim.height * sim.width, 5 * P + 16
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    pool = [torch.randint(0, 6, (P, n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(args.pool)]
    obs = sim.observation_world_major_tensor().to_torch()

    fence_token = torch.zeros(1, device="cuda") if use_dist and not rehearse else None

    def fence():
        """barrier + torch.cuda.synchronize().  Over RCCL the barrier is a one-element all_reduce enqueued behind the steps and waited
        for by the synchronize -- the rendezvous dist.barrier() makes, without ProcessGroupNCCL::barrier's own host-side waits
        (22 against 31 us per fence on an idle GPU, tools/fence_cost.py; the trailing fence is inside the timed block)."""
        if fence_token is not None:
            dist.all_reduce(fence_token)
        elif use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(x):
        if not use_dist:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        out = [torch.zeros_like(t) for _ in range(world_size)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def timed_blocks(one_step, steps, warmup, max_blocks=41):
        """`warmup` untimed steps, then blocks of exactly `steps` steps, each between two fences; every
        rank runs the same number of blocks (decided from the max-over-ranks time of the first)."""
        for i in range(warmup):
            one_step(i)
        blocks, mine = [], []
        count = 1
        k = 0
        while k < count:
            fence()
            t0 = time.perf_counter()
            for i in range(steps):
                one_step(i)
            fence()
            dt = time.perf_counter() - t0
            mine.append(dt)
            blocks.append(max_over_ranks(dt))
            if k == 0 and blocks[0] < MIN_BLOCK_SECONDS:
                count = int(min(max_blocks, max(3, MIN_BLOCK_SECONDS / max(blocks[0], 1e-6)))) | 1  # odd: a true median
            k += 1
        return blocks, mine

    # ---------------- headline: one launch per step, no communication ----------------
    blocks, mine = timed_blocks(lambda i: sim.step_with_actions(pool[i % args.pool]), args.steps, args.warmup)
    dt = statistics.median(blocks)
    # what a block of NO steps costs between the same fences: the share of a short block (--steps 20) that is the contract's
    # bracket, not the steps -- reported, never subtracted
    empty_blocks, _ = timed_blocks(lambda i: None, 0, 0, max_blocks=9)
    empty_block_ms = statistics.median(empty_blocks) * 1e3
    per_rank_ms = all_ranks(statistics.median(mine) / args.steps * 1e3)

    # ---------------- N > 1: the step followed by the all-gather of the observation shards ----------------
    gather = None
    if use_dist and not args.no_gather_leg:
        gathered = torch.empty((world_size * n,) + tuple(obs.shape[1:]), dtype=obs.dtype, device=obs.device)
        gsteps, gwarm = (min(args.steps, 3), 1) if rehearse else (args.steps, min(args.warmup, 10))

        def gather_step(i):
            sim.step_with_actions(pool[i % args.pool])
            gather_worlds(obs, 0, out=gathered)  # one all_gather_into_tensor straight into the global (N, P, H, W, F) tensor
        gblocks, gmine = timed_blocks(gather_step, gsteps, gwarm, max_blocks=9)
        gdt = statistics.median(gblocks)
        shard_bytes = obs.numel()
        step_s, gstep_s = dt / args.steps, gdt / gsteps
        gather = {"value": n * world_size * gsteps / gdt, "unit": "env-steps/s", "ms_per_step": gstep_s * 1e3, "steps": gsteps,
                  "bytes_per_rank_per_step": shard_bytes, "gathered_bytes_per_rank_per_step": shard_bytes * world_size,
                  "collective": "gloo all-gather staged through the host (rehearsal)" if rehearse
                  else "all_gather_into_tensor of the world-major int8 slab (RCCL)",
                  # what the collective adds to a step, and its bus bandwidth (every rank receives (G-1) shards)
                  "gather_ms_per_step": (gstep_s - step_s) * 1e3,
                  "busbw_GBps": shard_bytes * (world_size - 1) / (gstep_s - step_s) / 1e9 if gstep_s > step_s else None,
                  "per_rank_ms_per_step": all_ranks(statistics.median(gmine) / gsteps * 1e3)}

    # ---------------- roofline of the step kernel ----------------
    # average launch duration: HIP events on the launch stream (torch's current stream is the stream
    # the C ABI is handed) around back-to-back launches.  The queue never drains, so elapsed / launches
    # is the kernel's duration (rocprofv3 --kernel-trace agrees within 1%, profiles/); bracketing every
    # launch with its own event pair would add ~2 us of marker overhead to a ~10 us kernel.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def launches_us(fn, launches):
        torch.cuda.synchronize()
        ev0.record()
        for i in range(launches):
            fn(i)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3 / launches

    k_launch = max(args.steps, MIN_ROOFLINE_LAUNCHES)
    kernel_us = launches_us(lambda i: sim.step_with_actions(pool[i % args.pool]), k_launch)

    extras = {}
    solo = world_size == 1 and not use_dist
    single = solo and not args.no_extras
    stream = torch.cuda.current_stream().cuda_stream
    peak_measured = None
    if solo:
        # on-box bandwidth probes (second denominator): float4 copy and write-only streams over 1 GiB
        nbytes = 1 << 30
        a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        b = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        L = _lib.lib()

        def probe(mode, reps=8):
            _lib.check(L.mrl_probe_stream(a.data_ptr(), b.data_ptr(), nbytes, mode, local_rank, stream))
            us = launches_us(lambda i: _lib.check(L.mrl_probe_stream(a.data_ptr(), b.data_ptr(), nbytes, mode, local_rank, stream)), reps)
            return nbytes * (2 if mode == 0 else 1) / (us * 1e-6) / 1e9
        peak_measured = {"copy_GBps": probe(0), "write_GBps": probe(1), "write_through_GBps": probe(2),
                         "note": "float4 streams over 1 GiB (mrl_probe_stream): copy counts read + written bytes"}
        del a, b

    if single and args.fused_steps > 0:
        # the same random-policy workload with the actions drawn in the kernel and the worlds' state kept
        # in LDS between steps (mrl_rollout_random): no state or action bytes
        sim.rollout_random(args.fused_steps, seed=99, first_step=0)
        us = launches_us(lambda i: sim.rollout_random(args.fused_steps, seed=99, first_step=args.fused_steps), 1) / args.fused_steps
        bpw = P * C * F + 4 * P + 4
        extras["fused_random_rollout"] = {
            "value": n / (us * 1e-6), "unit": "env-steps/s", "steps_per_launch": args.fused_steps, "us_per_step": us,
            "bytes_per_world_step": bpw, "achieved_GBps": bpw * n / (us * 1e-6) / 1e9, "frac": bpw * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "note": "every step still writes its observation slab, rewards and dones; state stays in LDS, actions are drawn in-kernel"}
        # the SAME workload as the headline (the resident action pool, in order) as one open-loop sequence per launch
        k_seq = min(args.fused_steps, 512)
        seq_actions = torch.stack([pool[i % args.pool] for i in range(k_seq)]).contiguous()
        sim.step_sequence(seq_actions)
        us = launches_us(lambda i: sim.step_sequence(seq_actions), 1) / k_seq
        bpw = P * C * F + 4 * P + 4 + 4 * P
        extras["action_sequence_per_launch"] = {
            "value": n / (us * 1e-6), "unit": "env-steps/s", "steps_per_launch": k_seq, "us_per_step": us,
            "bytes_per_world_step": bpw, "achieved_GBps": bpw * n / (us * 1e-6) / 1e9, "frac": bpw * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "note": "same resident action pool, one launch per sequence; every step still writes observations, rewards, dones"}
        del seq_actions

    if single:
        # the headline loop once more as a captured HIP graph (the C ABI only enqueues on the caller's stream, so a step
        # call can be captured like any other stream work): the same launches, one per step, without the host's per-call
        # work -- what is left is the GPU's own time per step
        per_graph = 200
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for i in range(per_graph):
                    sim.step_with_actions(pool[i % args.pool])
        graph.replay()
        us = launches_us(lambda i: graph.replay(), 10) / per_graph
        extras["graph_replay"] = {
            "value": n / (us * 1e-6), "unit": "env-steps/s", "us_per_step": us, "steps_per_graph": per_graph,
            "frac": sim.bytes_per_world_step * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "note": "the headline's launches (one mrl_step_with_actions per step) captured once with torch.cuda.graph and replayed"}
        del graph

    if single:
        # the reference's own timed regions, on this engine's drop-in wrappers
        from madrona_rl_envs_playground_amd.envs import OvercookedMadrona
        env = OvercookedMadrona(args.layout, n, local_rank, horizon=args.horizon)
        acts64 = [p.to(torch.int64) for p in pool[:8]]  # the harness hands int64 actions (randint_like of a long tensor)
        reps = max(args.steps, MIN_ROOFLINE_LAUNCHES)
        for i in range(5):
            env.n_step(acts64[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            env.n_step(acts64[i % 8])
        torch.cuda.synchronize()
        w_dt = (time.perf_counter() - t0) / reps
        extras["wrapped_n_step"] = {"value": n / w_dt, "unit": "env-steps/s", "us_per_step": w_dt * 1e6, "steps": reps,
                                    "note": "OvercookedMadrona.n_step(actions int64 (P,N,1)): scripts/overcooked_example.py:104-106"}
        chosen = torch.zeros_like(env.static_actions)
        g_dones, g_obs, g_rew = (torch.zeros_like(env.static_dones), torch.zeros_like(env.static_observations),
                                 torch.zeros_like(env.static_rewards))

        def isolated(i):
            env.static_actions.copy_(chosen)
            env.sim.step()
            g_dones.copy_(env.static_dones)
            g_obs.copy_(env.static_observations)
            g_rew.copy_(env.static_rewards)
        for i in range(5):
            isolated(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            isolated(i)
        torch.cuda.synchronize()
        i_dt = (time.perf_counter() - t0) / reps
        extras["isolated_with_copies"] = {"value": n / i_dt, "unit": "env-steps/s", "us_per_step": i_dt * 1e6, "steps": reps,
                                          "note": "action copy + sim.step() + copies of done/obs/reward: scripts/overcooked_isolated_example.py:56-65"}
        del g_obs
        env.close()

    if single and not args.no_other_configs:
        extras["other_configs"] = other_configs(args, torch, local_rank, launches_us)

    large = None
    if single and args.large_worlds > n:
        # a launch far larger than the 256 MiB Infinity Cache: the kernel, not launch latency or cache residency
        big_n = args.large_worlds
        big = OvercookedSimulator(exec_mode=ExecMode.CUDA, gpu_id=local_rank, num_worlds=big_n, **params)
        big_pool = [torch.randint(0, 6, (P, big_n, 1), dtype=torch.int32, device="cuda", generator=gen) for _ in range(4)]
        for i in range(5):
            big.step_with_actions(big_pool[i % 4])
        us = launches_us(lambda i: big.step_with_actions(big_pool[i % 4]), 60)
        gbps = big.bytes_per_world_step * big_n / (us * 1e-6) / 1e9
        large = {"worlds": big_n, "kernel_us_avg": us, "value": big_n / (us * 1e-6), "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS,
                 "frac_of_measured_write_through": gbps / peak_measured["write_through_GBps"] if peak_measured else None}
        big.close()
        del big_pool

    if rank == 0:
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this
        # process): quoted only for this workload, this kernel and this build of csrc/
        traffic = measured_traffic(sim.kernel_name, f"overcooked {args.layout} {n}")
        bytes_per_launch = sim.bytes_per_world_step * n
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                    "kernel": sim.kernel_name, "kernel_us_avg": kernel_us, "bytes_per_world_step": sim.bytes_per_world_step,
                    "bytes_per_launch": bytes_per_launch, "launches_timed": k_launch,
                    "note": "peak = HBM spec; the 41 MB working set is rewritten in place every step and is smaller than the "
                            "256 MiB Infinity Cache, so the spec peak is a nominal denominator here: see peak_measured and large_batch"}
        if peak_measured:
            roofline["peak_measured"] = peak_measured
            roofline["frac_of_measured_write_through"] = achieved / peak_measured["write_through_GBps"]
        if large:
            roofline["large_batch"] = large
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
                       "worlds_per_gpu": n, "obs_gather": False},
            "timing": {"blocks": len(blocks), "block_ms": [b * 1e3 for b in blocks], "reported": "median block", "empty_block_ms": empty_block_ms,
                       "per_rank_ms_per_step": per_rank_ms, "rank_min_ms_per_step": min(per_rank_ms), "rank_max_ms_per_step": max(per_rank_ms)},
            "roofline": roofline,
            "build_hash": _lib.build_hash(),  # mrl_build_hash(): the sources the running library was compiled from; `traffic` is keyed by it
        }
        if use_dist:
            out["ranks"] = {"world_size": dist.get_world_size(), "backend": backend,
                            "rehearsal_on_one_gpu": bool(rehearse)}
            if gather is not None:
                gather["expected"] = expected_gather(obs.numel(), world_size, dt / args.steps)
                out["obs_gather"] = gather
            out["expected"] = {"value": [world_size * n / (x * 1e-3) for x in (9.0e-3, 8.0e-3)], "unit": "env-steps/s",
                               "basis": "no collective in the step: G x the single-GPU rate (8.0-9.0 us per step measured at N = 1, "
                                        "BENCH_r03 / profiles/r04_*), within the per-rank spread; DESIGN.md section 6"}
        out.update(extras)
        if not args.no_cpu_baseline and solo:
            out["cpu_baseline"] = cpu_baseline(params, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    sim.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if (args.gpus > 1 or force_dist()) and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run(args)


if __name__ == "__main__":
    main()
