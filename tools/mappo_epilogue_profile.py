#!/usr/bin/env python3
"""SURVEY.md section 8(f) item 3, measured: what the "rollout-buffer side" of the MAPPO loop costs on this engine.

The reference's loop (train/MAPPO/main_player.py:211-261) takes the int8 (N, W, H, F) observation the env returns,
casts it to float (r_actor_critic.py:57), views it channels-first for the CNN (utils/cnn.py:41 `movedim(-1,-3)`),
and copies obs/state into the rollout buffer (utils/shared_buffer.py:115 `chooseinsert`).  A fused epilogue of the
step kernel could write the float, channels-first slot directly.  This tool times the pieces with HIP events
(median of `--reps` repetitions each, 32768 worlds) so the decision rests on numbers:

  env_step          env.n_step (one fused kernel launch + the int64 -> int32 action copy)
  cast              obs.float()                       int8 (N,W,H,F) -> fp32, what r_actor_critic.py:57 does
  cast_movedim_conv the CNN's first layer on the cast, movedim'ed tensor (channels-last strides, no copy)
  policy_forward    whole actor + critic forward for one player
  insert_int8       buffer[t].copy_(obs)              what chooseinsert moves per player and step
  insert_fp32       the same slot kept in fp32        what a fused float epilogue would have to write
  loop              one full iteration of tools/mappo_rollout_loop.py
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mappo_rollout_loop as loop  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=32768)
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    n = args.worlds
    env, ego, buffers = loop.build(args.layout, n, steps_in_buffer=8)
    ob = env.reset()
    ob = loop.rollout(env, ego, buffers, ob, 10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        samples = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e3)
        return statistics.median(samples)

    acts = torch.randint(0, 6, (env.num_players, n, 1), device="cuda")
    obs = ob.obs
    as_float = obs.float()
    fp32_slot = torch.empty_like(as_float)
    out = {"workload": f"{args.layout}, {n} worlds, 1 GPU; microseconds, median of {args.reps}",
           "env_step": timed(lambda: env.n_step(acts)),
           "cast": timed(lambda: obs.float()),
           "cast_movedim_conv": timed(lambda: ego.actor.conv(obs.float().movedim(-1, -3))),
           "policy_forward": timed(lambda: ego.get_action(ob)),
           "insert_int8": timed(lambda: buffers["obs"][0].copy_(obs)),
           "insert_fp32": timed(lambda: fp32_slot.copy_(as_float))}
    state = {"ob": ob}

    def one_iteration():
        state["ob"] = loop.rollout(env, ego, buffers, state["ob"], 1)
    out["loop"] = timed(one_iteration)
    out["obs_bytes_int8"] = obs.numel()
    out["share_of_loop"] = {k: out[k] / out["loop"] for k in ("env_step", "cast", "insert_int8")}
    print(json.dumps(out))
    env.close()


if __name__ == "__main__":
    main()
