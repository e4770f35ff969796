"""MLP baseline (mirror of keisei/training/models/mlp.py:13-54): flatten -> [Linear, LayerNorm, ReLU]* ->
policy Linear(11259) / tanh value.  Scalar contract; BASELINE config 1 runs it on CPU only."""
from __future__ import annotations

import dataclasses

import torch
from torch import nn

from .base import BaseModel


@dataclasses.dataclass(frozen=True)
class MLPParams:
    hidden_sizes: list[int]

    def __post_init__(self) -> None:
        if any(s <= 0 for s in self.hidden_sizes):
            raise ValueError(f"All hidden_sizes must be > 0, got {self.hidden_sizes}")


class MLPModel(BaseModel):
    def __init__(self, params: MLPParams) -> None:
        super().__init__()
        width = self.OBS_CHANNELS * self.BOARD_SIZE * self.BOARD_SIZE
        stack: list[nn.Module] = []
        for size in params.hidden_sizes:
            stack += [nn.Linear(width, size), nn.LayerNorm(size), nn.ReLU()]
            width = size
        self.trunk = nn.Sequential(*stack)
        self.policy_fc = nn.Linear(width, self.ACTION_SPACE)
        self.value_fc = nn.Linear(width, 1)

    def forward(self, obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        self._check_obs(obs)
        h = self.trunk(obs.flatten(1))
        return self.policy_fc(h), torch.tanh(self.value_fc(h))
