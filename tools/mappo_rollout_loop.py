#!/usr/bin/env python3
"""Config 5 of BASELINE.json on one GPU: the MAPPO rollout inner loop around the env step.

The reference's loop (train/MAPPO/main_player.py:211-261, MainPlayer.next_step): actor + critic
forward for the ego and for the partner (a CentralizedAgent running the same networks,
train/partner_agents.py:27-62; CNN of train/MAPPO/utils/cnn.py:26-42: movedim(-1,-3), Conv2d 3x3,
two Linear layers), `envs.step`, `.clone()` of obs/state, insert into the rollout buffer.
The trainer itself is out of scope; this script only shows the engine
inside that loop with plain PyTorch networks of the same shape and reports env-steps/s of the loop
next to the bare env step.  Policy data-parallel consumers read the rank-local observation
views directly: no gather.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_rl_envs_playground_amd.envs import OvercookedMadrona  # noqa: E402
from madrona_rl_envs_playground_amd.pantheonrl_extension import VectorAgent  # noqa: E402


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
        if self.keep_inputs:  # obs.obs is a view of the simulator's buffer, which the next step overwrites
            self.last_obs = obs.obs.clone()
        self.last_action = torch.distributions.Categorical(logits=logits).sample().unsqueeze(-1)
        return self.last_action

    def update(self, rewards, dones):
        return None


def build(layout, n, gpu_id=0, horizon=400, steps_in_buffer=200, seed=0):
    """The env, the two policy-driven players and a rollout buffer, wired as the reference's
    MainPlayer does (train/MAPPO/main_player.py:73-112); returns (env, ego, buffers)."""
    torch.manual_seed(seed)
    env = OvercookedMadrona(layout, n, gpu_id, horizon=horizon)
    w, h, f = env.width, env.height, 5 * env.num_players + 16
    actor, critic = CNNBase(w, h, f).cuda(), CNNBase(w, h, f, out=1).cuda()
    ego = PolicyAgent(actor, critic)
    for _ in range(env.num_players - 1):
        env.add_partner_agent(PolicyAgent(actor, critic))
    T = steps_in_buffer
    buffers = {"obs": torch.empty((T, n, w, h, f), dtype=torch.int8, device="cuda"),
               "rew": torch.empty((T, n), dtype=torch.int32, device="cuda"),
               "done": torch.empty((T, n), dtype=torch.int32, device="cuda")}
    return env, ego, buffers


def rollout(env, ego, buffers, ob, steps, on_step=None):
    """`steps` iterations of MainPlayer.next_step (main_player.py:211-261): policy forward for the ego
    (the partner's runs inside env.step), env.step, clone into the buffer slot.  `on_step(t, ob_in,
    ego_action, ob_out, rew, done)` lets a test look at what the policy received."""
    T = buffers["obs"].shape[0]
    for t in range(steps):
        act = ego.get_action(ob)
        nxt, rew, done, _ = env.step(act)
        buffers["obs"][t % T].copy_(nxt.obs)          # the reference clones obs/state, then chooseinsert()s
        buffers["rew"][t % T].copy_(rew)
        buffers["done"][t % T].copy_(done)
        if on_step is not None:
            on_step(t, ob, act, nxt, rew, done)
        ob = nxt
    return ob


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--layout", default="cramped_room")
    args = ap.parse_args()
    n, T = args.worlds, args.steps
    env, ego, buffers = build(args.layout, n, steps_in_buffer=T)
    ob = env.reset()
    ob = rollout(env, ego, buffers, ob, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ob = rollout(env, ego, buffers, ob, T)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rand = torch.randint(0, 6, (env.num_players, n, 1), device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(T):
        env.n_step(rand)
    torch.cuda.synchronize()
    dt_env = time.perf_counter() - t1
    print(json.dumps({"workload": f"MAPPO-style rollout loop, {args.layout}, {n} worlds, 1 GPU",
                      "loop_env_steps_per_s": n * T / dt, "loop_ms_per_step": dt / T * 1e3,
                      "env_n_step_only_steps_per_s": n * T / dt_env, "env_n_step_only_ms": dt_env / T * 1e3}))
    env.close()


if __name__ == "__main__":
    main()
