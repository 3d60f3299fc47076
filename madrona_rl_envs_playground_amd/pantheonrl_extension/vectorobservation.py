"""``VectorObservation`` -- same fields and defaulting rule as the reference's
(/root/reference/pantheonrl_extension/vectorobservation.py:5-32)."""
from typing import Optional

import torch


class VectorObservation:
    """Batched observation of one agent over N parallel worlds.

    active       (N,) bool   which worlds expect an action from this agent
    obs          (N, *obs_shape)
    state        (N, *state_shape); defaults to ``obs``
    action_mask  (N, num_actions) bool or None (= everything legal)
    """

    __slots__ = ("active", "obs", "state", "action_mask")

    def __init__(self, active: torch.Tensor, obs: torch.Tensor, state: Optional[torch.Tensor] = None,
                 action_mask: Optional[torch.Tensor] = None):
        self.active = active
        self.obs = obs
        self.state = obs if state is None else state
        self.action_mask = action_mask

    def __repr__(self):
        return (f"VectorObservation(active={tuple(self.active.shape)}, obs={tuple(self.obs.shape)}, "
                f"state={tuple(self.state.shape)}, action_mask="
                f"{None if self.action_mask is None else tuple(self.action_mask.shape)})")
