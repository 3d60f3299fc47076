"""Several Overcooked layouts stepped as ONE batch (SURVEY.md section 8(f) item 2's extension).

The reference's ``Config`` holds a single terrain (src/overcooked_env/sim.hpp:44-57), so one of its simulators is
one layout; a curriculum over the five standard layouts needs five simulators stepped one after the other.  Layouts
differ in height x width (and so in observation shape), which rules out one tensor for all of them; what CAN be
shared is the device time: ``OvercookedMultiLayout`` keeps one simulator per layout and enqueues their steps on
separate HIP streams, forked from and joined back into the caller's stream with events, so the kernels of the
sub-batches run side by side and a small sub-batch does not leave the GPU idle.  Every sub-batch behaves exactly like
its own ``OvercookedMadrona`` (same tensors, same values: tests compare them).

``graph=True``: the whole fork / step / join pattern -- one launch per layout on its own stream, two event operations
each -- is captured ONCE as a HIP graph and every later ``n_step`` replays it: one host call per step for all layouts
instead of one step call and two event calls per layout (Overcooked step launches carry no host-side state, so they
can be captured: INTEGRATION.md).  The captured launches read the simulators' ACTION tensors, so the caller's actions
are copied there first (or written there in place by the caller, ``n_step(None)``, as the reference's raw loops do).
"""
import torch

from .overcooked_env import OvercookedMadrona


class OvercookedMultiLayout:
    """``layouts``: list of layout names (or ``.layout`` paths); ``num_envs``: an int (per layout) or a list."""

    def __init__(self, layouts, num_envs, gpu_id, horizon=400, num_players=None, graph=False):
        counts = [int(num_envs)] * len(layouts) if isinstance(num_envs, int) else [int(n) for n in num_envs]
        if len(counts) != len(layouts):
            raise ValueError("one world count per layout")
        self.layouts = list(layouts)
        self.envs = [OvercookedMadrona(name, n, gpu_id, horizon=horizon, num_players=num_players)
                     for name, n in zip(self.layouts, counts)]
        self.num_envs = sum(counts)
        self.device = self.envs[0].device
        self._streams = [torch.cuda.Stream(device=self.device) for _ in self.envs]
        self._fork = torch.cuda.Event()
        self._joins = [torch.cuda.Event() for _ in self.envs]
        self._use_graph = bool(graph)
        self._graph = None

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

    def n_step(self, actions):
        """``actions``: one (P_i, N_i, 1) tensor per layout -> list of (obs, rewards, dones, infos), one per layout.
        The results are valid on the caller's current stream when this returns (as for a single env)."""
        if actions is not None and len(actions) != len(self.envs):
            raise ValueError("one action tensor per layout")
        if self._use_graph:
            if actions is not None:
                for env, act in zip(self.envs, actions):
                    env.static_actions.copy_(act.to(env.static_actions.device), non_blocking=True)
            if self._graph is None:
                self._capture()  # (capturing does not execute: the replay below is this call's step)
            self._graph.replay()
            return [(env.get_obs(), env.static_rewards, env.static_dones, env.infos) for env in self.envs]
        if actions is None:
            actions = [env.static_actions for env in self.envs]
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
