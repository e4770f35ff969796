"""SE-ResNet with KataGo-style global-pool bias blocks and three heads.

API mirror of keisei/training/models/se_resnet.py:15-159 -- same class names, constructor
arguments, child-module names (hence identical ``state_dict`` keys/shapes) and
error behaviour.  Parameters are stored exactly as the reference stores them (fp32, torch
layouts); kernel-friendly copies are derived caches inside the HIP engine.

Dispatch: tensors on a CUDA/HIP device run on the hand-written HIP kernels
(keisei_amd/hip/seresnet.py -> libkeisei_amd.so) and raise if that library is unavailable --
there is no PyTorch fallback on the GPU.  CPU tensors (the reference's whole test-suite runs on
CPU) use the plain ``nn`` children below, which is what makes this an ordinary ``nn.Module``.
"""
from __future__ import annotations

import dataclasses

import torch
import torch.nn.functional as F
from torch import nn

from .katago_base import KataGoBaseModel, KataGoOutput

_POSITIVE_FIELDS = ("num_blocks", "channels", "se_reduction", "global_pool_channels", "policy_channels",
                    "value_fc_size", "score_fc_size", "obs_channels")


@dataclasses.dataclass(frozen=True)
class SEResNetParams:
    num_blocks: int = 40
    channels: int = 256
    se_reduction: int = 16
    global_pool_channels: int = 128
    policy_channels: int = 32
    value_fc_size: int = 256
    score_fc_size: int = 128
    obs_channels: int = 50

    def __post_init__(self) -> None:
        for name in _POSITIVE_FIELDS:
            value = getattr(self, name)
            if value < 1:
                raise ValueError(f"{name} must be >= 1, got {value}")
        if self.channels // self.se_reduction < 1:
            raise ValueError(f"channels ({self.channels}) // se_reduction ({self.se_reduction}) must be >= 1")


def _global_pool(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) -> (B,3C): spatial mean, max and population std, concatenated."""
    spatial = (-2, -1)
    return torch.cat((x.mean(dim=spatial), x.amax(dim=spatial), x.std(dim=spatial, correction=0)), dim=-1)


class GlobalPoolBiasBlock(nn.Module):
    """conv1-BN-ReLU (+ bias from the pooled block input) - conv2-BN - SE scale/shift - residual - ReLU."""

    def __init__(self, channels: int, se_reduction: int, global_pool_channels: int) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)
        self.global_fc = nn.Sequential(
            nn.Linear(3 * channels, global_pool_channels), nn.ReLU(), nn.Linear(global_pool_channels, channels))
        self.se_fc1 = nn.Linear(channels, channels // se_reduction)
        self.se_fc2 = nn.Linear(channels // se_reduction, 2 * channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda:
            from keisei_amd.hip.block import run_block
            return run_block(self, x)
        bias = self.global_fc(_global_pool(x))
        h = F.relu(self.bn1(self.conv1(x))) + bias[:, :, None, None]
        z = self.bn2(self.conv2(h))
        gate, shift = self.se_fc2(F.relu(self.se_fc1(z.mean(dim=(-2, -1))))).chunk(2, dim=-1)
        return F.relu(z * torch.sigmoid(gate)[:, :, None, None] + shift[:, :, None, None] + x)


class SEResNetModel(KataGoBaseModel):
    def __init__(self, params: SEResNetParams) -> None:
        super().__init__()
        self.params = params
        c = params.channels
        self.input_conv = nn.Conv2d(params.obs_channels, c, 3, padding=1, bias=False)
        self.input_bn = nn.BatchNorm2d(c)
        self.blocks = nn.Sequential(*(GlobalPoolBiasBlock(c, params.se_reduction, params.global_pool_channels)
                                      for _ in range(params.num_blocks)))
        self.policy_conv1 = nn.Conv2d(c, params.policy_channels, 1, bias=False)
        self.policy_bn1 = nn.BatchNorm2d(params.policy_channels)
        self.policy_conv2 = nn.Conv2d(params.policy_channels, self.SPATIAL_MOVE_TYPES, 1)
        self.value_fc1 = nn.Linear(3 * c, params.value_fc_size)
        self.value_fc2 = nn.Linear(params.value_fc_size, 3)
        self.score_fc1 = nn.Linear(3 * c, params.score_fc_size)
        self.score_fc2 = nn.Linear(params.score_fc_size, 1)

    def _check_obs(self, obs: torch.Tensor) -> None:
        c = self.params.obs_channels
        if obs.ndim != 4 or obs.shape[1] != c or obs.shape[2] != 9 or obs.shape[3] != 9:
            raise ValueError(f"Expected obs shape (batch, {c}, 9, 9), got {tuple(obs.shape)}")

    def _forward_impl(self, obs: torch.Tensor, gather_idx: torch.Tensor | None = None) -> KataGoOutput:
        self._check_obs(obs)
        if obs.is_cuda:
            from keisei_amd.hip.seresnet import run_model
            policy, value, score = run_model(self, obs, gather_idx)
            return KataGoOutput(policy_logits=policy, value_logits=value, score_lead=score)
        if gather_idx is not None:
            obs = obs[gather_idx]
        x = self.blocks(F.relu(self.input_bn(self.input_conv(obs))))
        policy = self.policy_conv2(F.relu(self.policy_bn1(self.policy_conv1(x)))).permute(0, 2, 3, 1)
        pooled = _global_pool(x)
        value = self.value_fc2(F.relu(self.value_fc1(pooled)))
        score = self.score_fc2(F.relu(self.score_fc1(pooled)))
        return KataGoOutput(policy_logits=policy, value_logits=value, score_lead=score)
