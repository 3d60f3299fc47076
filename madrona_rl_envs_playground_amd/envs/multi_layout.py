"""Several Overcooked layouts stepped as ONE batch (SURVEY.md section 8(f) item 2's extension).

The reference's ``Config`` holds a single terrain (src/overcooked_env/sim.hpp:44-57), so one of its simulators is
one layout; a curriculum over the five standard layouts needs five simulators stepped one after the other.  Layouts
differ in height x width (and so in observation shape), which rules out one tensor for all of them; what CAN be
shared is the device time: ``OvercookedMultiLayout`` keeps one simulator per layout and enqueues their steps on
separate HIP streams, forked from and joined back into the caller's stream with events, so the kernels of the
sub-batches run side by side and a small sub-batch does not leave the GPU idle.  Every sub-batch behaves exactly like
its own ``OvercookedMadrona`` (same tensors, same values: tests compare them).
"""
import torch

from .overcooked_env import OvercookedMadrona


class OvercookedMultiLayout:
    """``layouts``: list of layout names (or ``.layout`` paths); ``num_envs``: an int (per layout) or a list."""

    def __init__(self, layouts, num_envs, gpu_id, horizon=400, num_players=None):
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

    def n_reset(self):
        return [env.n_reset() for env in self.envs]

    def n_step(self, actions):
        """``actions``: one (P_i, N_i, 1) tensor per layout -> list of (obs, rewards, dones, infos), one per layout.
        The results are valid on the caller's current stream when this returns (as for a single env)."""
        if len(actions) != len(self.envs):
            raise ValueError("one action tensor per layout")
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
