#!/usr/bin/env python3
"""Config 5 of BASELINE.json: the MAPPO rollout inner loop around the env step, on 1..8 GPUs.

The reference's loop (train/MAPPO/main_player.py:211-261, MainPlayer.next_step): actor + critic
forward for the ego and for the partner (a CentralizedAgent running the same networks,
train/partner_agents.py:27-62; CNN of train/MAPPO/utils/cnn.py:26-42: movedim(-1,-3), Conv2d 3x3,
two Linear layers), `envs.step`, `.clone()` of obs/state, insert into the rollout buffer
(utils/shared_buffer.py:115).  The trainer itself is out of scope; this
script shows the engine inside that loop with plain PyTorch networks of the same shape and reports
env-steps/s of the loop next to the bare env step.

Rollout buffer: by default the step kernel writes each step's observations straight INTO the buffer
slot of that step (`env.step(act, out=slot)` -> mrl_set_observation_output): the slot is the kernel's
own world-major (N, P, H, W, F) int8 block, the per-player observations the policies read are views of
it, and the clone + insert of the reference's loop is gone (`--copy-insert` keeps the copy, for the
comparison).

    python tools/mappo_rollout_loop.py --gpus N
One rank per GPU (started here as a child `python -m torch.distributed.run ...` before anything
touches a GPU, like bench.py; under a launcher this process is one rank): every rank owns an env shard
of `--worlds` worlds and a replica of the policies (same seed, same weights).  Policy and env are
co-sharded, so the rollout loop has NO collective; the ranks meet at a barrier in front of and behind
the timed region, rank 0 gathers the per-rank times and prints ONE JSON line with per-rank and summed
env-steps/s.  MRL_BENCH_REHEARSE=1 puts all ranks on GPU 0 over gloo (one-GPU rehearsal of the rank
protocol); MRL_BENCH_FORCE_DIST=1 runs the protocol over nccl at world_size 1.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _torch_side():
    """Everything that needs torch, imported only in a rank process."""
    import torch
    import torch.nn as nn
    from madrona_rl_envs_playground_amd.pantheonrl_extension import VectorAgent

    class CNNBase(nn.Module):
        def __init__(self, w, h, f, hidden=64, out=6):
            super().__init__()
            self.conv = nn.Conv2d(f, 32, kernel_size=3, stride=1)
            self.fc = nn.Sequential(nn.Linear(32 * (w - 2) * (h - 2), hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU())
            self.head = nn.Linear(hidden, out)

        def forward(self, obs_i8):
            x = obs_i8.float().movedim(-1, -3)            # (N, W, H, F) int8 -> (N, F, W, H) float
            x = torch.relu(self.conv(x)).flatten(1)
            return self.head(self.fc(x))

    class PolicyAgent(VectorAgent):
        def __init__(self, actor, critic):
            self.actor, self.critic = actor, critic
            self.keep_inputs = False  # tests: keep a copy of every observation this policy is shown

        @torch.no_grad()
        def get_action(self, obs, record=True):
            logits = self.actor(obs.obs)
            self.value = self.critic(obs.state)
            if self.keep_inputs:  # obs.obs is a view of a buffer a later step overwrites
                self.last_obs = obs.obs.clone()
            self.last_action = torch.distributions.Categorical(logits=logits).sample().unsqueeze(-1)
            return self.last_action

        def update(self, rewards, dones):
            return None

    return torch, CNNBase, PolicyAgent


def build(layout, n, gpu_id=0, horizon=400, steps_in_buffer=200, seed=0, in_place=True):
    """The env, the two policy-driven players and a rollout buffer, wired as the reference's
    MainPlayer does (train/MAPPO/main_player.py:73-112); returns (env, ego, buffers).
    in_place: buffers["obs"] is a ring of world-major slots (T, N, P, H, W, F) the step kernel writes into;
    otherwise (T, N, W, H, F) copies of the ego's observation, as the reference keeps them."""
    torch, CNNBase, PolicyAgent = _torch_side()
    from madrona_rl_envs_playground_amd.envs import OvercookedMadrona
    torch.manual_seed(seed)
    dev = torch.device("cuda", gpu_id)
    env = OvercookedMadrona(layout, n, gpu_id, horizon=horizon)
    w, h, f, P = env.width, env.height, 5 * env.num_players + 16, env.num_players
    actor, critic = CNNBase(w, h, f).to(dev), CNNBase(w, h, f, out=1).to(dev)
    ego = PolicyAgent(actor, critic)
    for _ in range(env.num_players - 1):
        env.add_partner_agent(PolicyAgent(actor, critic))
    T = steps_in_buffer
    shape = (T, n, P, h, w, f) if in_place else (T, n, w, h, f)
    buffers = {"obs": torch.empty(shape, dtype=torch.int8, device=dev),
               "rew": torch.empty((T, n), dtype=torch.int32, device=dev),
               "done": torch.empty((T, n), dtype=torch.int32, device=dev),
               "in_place": in_place}
    return env, ego, buffers


def rollout(env, ego, buffers, ob, steps, on_step=None):
    """`steps` iterations of MainPlayer.next_step (main_player.py:211-261): policy forward for the ego
    (the partner's runs inside env.step), env.step, the step's observations into the buffer slot -- written
    there by the step kernel itself, or cloned + inserted like the reference.  `on_step(t, ob_in, ego_action,
    ob_out, rew, done)` lets a test look at what the policy received."""
    T = buffers["obs"].shape[0]
    for t in range(steps):
        act = ego.get_action(ob)
        if buffers["in_place"]:
            nxt, rew, done, _ = env.step(act, out=buffers["obs"][t % T])
        else:
            nxt, rew, done, _ = env.step(act)
            buffers["obs"][t % T].copy_(nxt.obs)      # the reference clones obs/state, then chooseinsert()s
        buffers["rew"][t % T].copy_(rew)
        buffers["done"][t % T].copy_(done)
        if on_step is not None:
            on_step(t, ob, act, nxt, rew, done)
        ob = nxt
    return ob


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--worlds", type=int, default=32768, help="worlds per GPU")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--copy-insert", action="store_true", help="clone + insert the observations like the reference instead of stepping into the slot")
    return ap.parse_args(argv)


def spawn_ranks(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def run(args):
    import torch
    import torch.distributed as dist
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
    rehearse = world_size > 1 and os.environ.get("MRL_BENCH_REHEARSE") == "1"
    use_dist = world_size > 1 or os.environ.get("MRL_BENCH_FORCE_DIST") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if rehearse else "nccl"
        sys.stdout.flush()
        keep = os.dup(1)  # gloo announces its connections and RCCL its version on stdout; stdout carries the JSON line only
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(x):
        if not use_dist:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        out = [torch.zeros_like(t) for _ in range(world_size)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    n, T = args.worlds, args.steps
    env, ego, buffers = build(args.layout, n, gpu_id=local_rank, steps_in_buffer=T, seed=0, in_place=not args.copy_insert)
    ob = env.reset()
    ob = rollout(env, ego, buffers, ob, 10)
    fence()
    t0 = time.perf_counter()
    ob = rollout(env, ego, buffers, ob, T)   # no collective inside: env shard and policy replica live on the same rank
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0            # behind the barrier: the slowest rank's time
    rand = torch.randint(0, 6, (env.num_players, n, 1), device="cuda")
    fence()
    t1 = time.perf_counter()
    for _ in range(T):
        env.n_step(rand)
    torch.cuda.synchronize()
    dt_env = time.perf_counter() - t1
    # the row's own piece without the policy in the way: env step + getting the ego's observation into the buffer, us per step
    ring = torch.empty((8,) + tuple(env.static_world_major_observations.shape), dtype=torch.int8, device="cuda")
    copies = torch.empty((8, n, env.width, env.height, 5 * env.num_players + 16), dtype=torch.int8, device="cuda")

    def timed(fn):
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(T):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / T * 1e6
    insert_us = {"step_into_slot": timed(lambda i: env.n_step(rand, out=ring[i % 8])),
                 "step_then_clone_insert": timed(lambda i: copies[i % 8].copy_(env.n_step(rand)[0][env.ego_ind].obs))}
    env.n_step(rand)  # hand the output back to the simulator's own tensor
    del ring, copies
    per_rank = all_ranks(mine)
    per_rank_env = all_ranks(dt_env)
    if rank == 0:
        slowest = max(per_rank + [dt]) if use_dist else mine
        out = {"workload": f"MAPPO-style rollout loop, {args.layout}, {n} worlds per GPU, {world_size} GPU(s)",
               "observations": "cloned + inserted (reference)" if args.copy_insert else "written into the buffer slot by the step kernel",
               "n_gpus": world_size, "steps": T,
               "loop_env_steps_per_s": n * world_size * T / slowest, "loop_ms_per_step": slowest / T * 1e3,
               "per_rank_loop_env_steps_per_s": [n * T / x for x in per_rank],
               "sum_of_ranks_env_steps_per_s": sum(n * T / x for x in per_rank),
               "env_n_step_only_steps_per_s": sum(n * T / x for x in per_rank_env), "env_n_step_only_ms": max(per_rank_env) / T * 1e3,
               "env_step_plus_buffer_insert_us": insert_us}
        if use_dist:
            out["ranks"] = {"world_size": dist.get_world_size(), "backend": backend, "rehearsal_on_one_gpu": bool(rehearse)}
        print(json.dumps(out), flush=True)
    env.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    force = os.environ.get("MRL_BENCH_FORCE_DIST") == "1"
    if (args.gpus > 1 or force) and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run(args)


if __name__ == "__main__":
    main()
