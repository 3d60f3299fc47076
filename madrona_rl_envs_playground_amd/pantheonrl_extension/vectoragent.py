"""Agent interface the vector envs drive their partner players through
(/root/reference/pantheonrl_extension/vectoragent.py:9-41)."""
from abc import ABC, abstractmethod

import torch

from .vectorobservation import VectorObservation


class VectorAgent(ABC):
    @abstractmethod
    def get_action(self, obs: VectorObservation, record: bool = True) -> torch.Tensor:
        """Actions for all N worlds given this agent's observation."""

    @abstractmethod
    def update(self, rewards: torch.Tensor, dones: torch.Tensor) -> None:
        """Reward / done feedback for the most recent recorded ``get_action``."""


class RandomVectorAgent(VectorAgent):
    """Calls ``sampler()`` for every action (the reference's random partner)."""

    def __init__(self, sampler):
        self.sampler = sampler

    def get_action(self, obs: VectorObservation, record: bool = True) -> torch.Tensor:
        return self.sampler()

    def update(self, rewards: torch.Tensor, dones: torch.Tensor) -> None:
        return None
