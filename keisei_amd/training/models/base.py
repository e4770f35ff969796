"""Scalar-contract base (mirror of keisei/training/models/base.py:11-27):
forward(obs (B,50,9,9)) -> (policy_logits (B,11259) raw, tanh value (B,1))."""
from __future__ import annotations

import abc

import torch
from torch import nn


class BaseModel(abc.ABC, nn.Module):
    OBS_CHANNELS = 50
    BOARD_SIZE = 9
    ACTION_SPACE = 11259

    @abc.abstractmethod
    def forward(self, obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]: ...

    def _check_obs(self, obs: torch.Tensor) -> None:
        want = (self.OBS_CHANNELS, self.BOARD_SIZE, self.BOARD_SIZE)
        if obs.ndim != 4 or tuple(obs.shape[1:]) != want:
            hint = " (input appears to be NHWC — expected NCHW)" if obs.ndim == 4 and obs.shape[-1] == want[0] else ""
            raise ValueError(f"Expected obs shape (batch, {want[0]}, {want[1]}, {want[2]}), got {tuple(obs.shape)}{hint}")
