"""Several Overcooked layouts stepped as ONE batch (SURVEY.md section 8(f) item 2's extension).

The reference's ``Config`` holds a single terrain (src/overcooked_env/sim.hpp:44-57), so one of its simulators is
one layout; a curriculum over the five standard layouts needs five simulators stepped one after the other.  Layouts
differ in height x width (and so in observation shape), which rules out one tensor for all of them; what CAN be
shared is the launch.  ``OvercookedMultiLayout`` keeps one simulator per layout -- every sub-batch behaves exactly like
its own ``OvercookedMadrona`` (same tensors, same values: tests compare them) -- and steps them, ``mode``:

``"one_launch"``   ONE kernel launch for all sub-batches (``mrl_step_many``: the grid is the concatenation of the simulators'
                   grids, every workgroup runs its own simulator's step on that simulator's parameters, which travel in the
                   kernel arguments).  One host call per step, no streams, no events.  It runs the generic step kernel, so
                   it is for many small sub-batches, where the launches are what costs.  A launch takes up to eight
                   simulators (more are stepped eight at a time), and a simulator ``mrl_step_many`` does not take -- a large
                   layout whose workgroups share one copy of a world (``simulators.can_step_with_others``) -- is stepped by
                   its own call behind the shared launch.
``"sequential"``   one step call per layout on the caller's stream, each with its own (specialised) kernel: best once the
                   sub-batches are large enough to fill the GPU on their own.
``"graph"``        the fork / step / join pattern below captured ONCE as a HIP graph and replayed per step (Overcooked step
                   launches carry no host-side state, so they can be captured: INTEGRATION.md); the captured launches read
                   the simulators' ACTION tensors, so the caller's actions are copied there first.
``"forked_streams"`` one stream per layout, forked from and joined back into the caller's stream with events.
``"auto"``         (default) ``one_launch`` while every sub-batch has at most 16384 worlds and at least two of them can share
                   a launch, ``sequential`` otherwise.

Measured, the five standard layouts, us per step for all five sub-batches (tools/multi_layout_probe.py):
5 x 100 worlds: one launch 11.2, sequential 25, graph 40, forked streams 110 (the event operations cost the host more than
the launches); 5 x 8192: 26.3 / 35.7 / 54 / 88; 5 x 32768: 75 / 61.5 / 80 / 130.
"""
import torch

from ..simulators import STEP_MANY_MAX, can_step_with_others, step_many
from .overcooked_env import OvercookedMadrona

ONE_LAUNCH_MAX_WORLDS = 16384
MODES = ("auto", "one_launch", "sequential", "graph", "forked_streams")


class OvercookedMultiLayout:
    """``layouts``: list of layout names (or ``.layout`` paths); ``num_envs``: an int (per layout) or a list."""

    def __init__(self, layouts, num_envs, gpu_id, horizon=400, num_players=None, mode="auto"):
        if mode not in MODES:
            raise ValueError(f"mode must be one of {MODES}")
        counts = [int(num_envs)] * len(layouts) if isinstance(num_envs, int) else [int(n) for n in num_envs]
        if len(counts) != len(layouts):
            raise ValueError("one world count per layout")
        self.layouts = list(layouts)
        self.envs = [OvercookedMadrona(name, n, gpu_id, horizon=horizon, num_players=num_players)
                     for name, n in zip(self.layouts, counts)]
        self.num_envs = sum(counts)
        self.device = self.envs[0].device
        # who can share a launch (mrl_step_many refuses the shared-world configuration), eight at a time; the others alone
        together = [k for k, env in enumerate(self.envs) if can_step_with_others(env.sim)]
        self._shared = [together[i:i + STEP_MANY_MAX] for i in range(0, len(together), STEP_MANY_MAX)]
        self._alone = [k for k in range(len(self.envs)) if k not in together]
        if mode == "auto":
            mode = "one_launch" if max(counts) <= ONE_LAUNCH_MAX_WORLDS and len(together) >= 2 else "sequential"
        self.mode = mode
        self._graph = None
        if mode in ("graph", "forked_streams"):
            self._streams = [torch.cuda.Stream(device=self.device) for _ in self.envs]
            self._fork = torch.cuda.Event()
            self._joins = [torch.cuda.Event() for _ in self.envs]

    def _capture(self):
        """fork -> every layout's step on its own stream -> join, as one graph (captured on a side stream)."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for env, stream in zip(self.envs, self._streams):
                    stream.wait_stream(side)
                    with torch.cuda.stream(stream):
                        env.sim.step()
                    side.wait_stream(stream)
        torch.cuda.current_stream(self.device).wait_stream(side)
        self._graph = graph

    def n_reset(self):
        return [env.n_reset() for env in self.envs]

    def _results(self):
        return [(env.get_obs(), env.static_rewards, env.static_dones, env.infos) for env in self.envs]

    def n_step(self, actions):
        """``actions``: one (P_i, N_i, 1) tensor per layout (or None: the caller has written the simulators' ACTION tensors
        in place, as the reference's raw loops do) -> list of (obs, rewards, dones, infos), one per layout.  The results are
        valid on the caller's current stream when this returns (as for a single env)."""
        if actions is not None and len(actions) != len(self.envs):
            raise ValueError("one action tensor per layout")
        if self.mode == "one_launch":
            ready = None
            if actions is not None:
                ready = []
                for env, act in zip(self.envs, actions):
                    if act.dtype == torch.int32 and act.is_cuda and act.is_contiguous() and act.shape == env.static_actions.shape:
                        ready.append(act)          # read where it is
                    else:
                        env.static_actions.copy_(act.to(env.static_actions.device), non_blocking=True)
                        ready.append(None)         # the simulator's own ACTION tensor
            for group in self._shared:
                step_many([self.envs[k].sim for k in group], None if ready is None else [ready[k] for k in group])
            for k in self._alone:
                if ready is None or ready[k] is None:
                    self.envs[k].sim.step()
                else:
                    self.envs[k].sim.step_with_actions(ready[k])
            return self._results()
        if self.mode == "graph":
            if actions is not None:
                for env, act in zip(self.envs, actions):
                    env.static_actions.copy_(act.to(env.static_actions.device), non_blocking=True)
            if self._graph is None:
                self._capture()  # (capturing does not execute: the replay below is this call's step)
            self._graph.replay()
            return self._results()
        if actions is None:
            actions = [env.static_actions for env in self.envs]
        if self.mode == "sequential":
            out = []
            for env, act in zip(self.envs, actions):
                if act.dtype == torch.int32 and act.is_cuda and act.is_contiguous() and act.shape == env.static_actions.shape:
                    env.sim.step_with_actions(act)  # read where it is, like one_launch (env.n_step would copy it into static_actions first)
                    out.append((env.get_obs(), env.static_rewards, env.static_dones, env.infos))
                else:
                    out.append(env.n_step(act))
            return out
        caller = torch.cuda.current_stream(self.device)
        self._fork.record(caller)
        out = []
        for env, act, stream, join in zip(self.envs, actions, self._streams, self._joins):
            stream.wait_event(self._fork)          # the actions were produced on the caller's stream
            with torch.cuda.stream(stream):
                out.append(env.n_step(act))
                join.record(stream)
        for join in self._joins:
            caller.wait_event(join)
        return out

    def close(self):
        for env in self.envs:
            env.close()
