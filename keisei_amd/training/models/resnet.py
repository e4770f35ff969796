"""Plain ResNet, scalar contract (mirror of keisei/training/models/resnet.py:13-83).  Kept for registry
completeness (``build_model("resnet", ...)``); no reference config trains it, so it has no HIP path."""
from __future__ import annotations

import dataclasses

import torch
from torch import nn

from .base import BaseModel


@dataclasses.dataclass(frozen=True)
class ResNetParams:
    hidden_size: int
    num_layers: int

    def __post_init__(self) -> None:
        if self.hidden_size <= 0:
            raise ValueError(f"hidden_size must be > 0, got {self.hidden_size}")
        if self.num_layers < 0:
            raise ValueError(f"num_layers must be >= 0, got {self.num_layers}")


class ResidualBlock(nn.Module):
    def __init__(self, channels: int) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = torch.relu(self.bn1(self.conv1(x)))
        return torch.relu(self.bn2(self.conv2(h)) + x)


class ResNetModel(BaseModel):
    def __init__(self, params: ResNetParams) -> None:
        super().__init__()
        c, sq = params.hidden_size, self.BOARD_SIZE * self.BOARD_SIZE
        self.input_conv = nn.Conv2d(self.OBS_CHANNELS, c, 3, padding=1, bias=False)
        self.input_bn = nn.BatchNorm2d(c)
        self.blocks = nn.Sequential(*(ResidualBlock(c) for _ in range(params.num_layers)))
        self.policy_conv = nn.Conv2d(c, 2, 1, bias=False)
        self.policy_bn = nn.BatchNorm2d(2)
        self.policy_fc = nn.Linear(2 * sq, self.ACTION_SPACE)
        self.value_conv = nn.Conv2d(c, 1, 1, bias=False)
        self.value_bn = nn.BatchNorm2d(1)
        self.value_fc1 = nn.Linear(sq, c)
        self.value_fc2 = nn.Linear(c, 1)

    def forward(self, obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        self._check_obs(obs)
        x = self.blocks(torch.relu(self.input_bn(self.input_conv(obs))))
        policy = self.policy_fc(torch.relu(self.policy_bn(self.policy_conv(x))).flatten(1))
        v = torch.relu(self.value_fc1(torch.relu(self.value_bn(self.value_conv(x))).flatten(1)))
        return policy, torch.tanh(self.value_fc2(v))
