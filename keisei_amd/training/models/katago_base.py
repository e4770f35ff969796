"""KataGo-style multi-head model contract.

Mirrors keisei/training/models/katago_base.py:14-78: the three-field output container and the
abstract base whose ``forward`` optionally wraps ``_forward_impl`` in ``torch.amp.autocast``
according to ``configure_amp`` (which refuses changes once ``_amp_frozen`` is set).
"""
from __future__ import annotations

import abc
import dataclasses

import torch
from torch import nn


@dataclasses.dataclass
class KataGoOutput:
    """policy_logits (B,9,9,139) raw/unmasked; value_logits (B,3) W/D/L; score_lead (B,1)."""

    policy_logits: torch.Tensor
    value_logits: torch.Tensor
    score_lead: torch.Tensor


class KataGoBaseModel(abc.ABC, nn.Module):
    BOARD_SIZE = 9
    SPATIAL_MOVE_TYPES = 139
    SPATIAL_ACTION_SPACE = 81 * 139

    def __init__(self) -> None:
        super().__init__()
        self._amp_enabled: bool = False
        self._amp_dtype: torch.dtype = torch.float16
        self._amp_device_type: str = "cpu"
        self._amp_frozen: bool = False

    def configure_amp(self, enabled: bool, dtype: torch.dtype = torch.float16, device_type: str = "cuda") -> None:
        if self._amp_frozen:
            raise RuntimeError(
                "configure_amp() must not be called after torch.compile() — "
                "changing AMP attributes would trigger silent recompilation"
            )
        self._amp_enabled, self._amp_dtype, self._amp_device_type = enabled, dtype, device_type

    def forward(self, obs: torch.Tensor, gather_idx: torch.Tensor | None = None) -> KataGoOutput:
        """``gather_idx`` (extension, optional): evaluate ``obs[gather_idx]`` with the row gather fused
        into the first kernel -- used by the PPO minibatch loop on a device-resident epoch dataset."""
        extra = () if gather_idx is None else (gather_idx,)
        if self._amp_enabled:
            with torch.amp.autocast(device_type=self._amp_device_type, dtype=self._amp_dtype):
                return self._forward_impl(obs, *extra)
        return self._forward_impl(obs, *extra)

    @abc.abstractmethod
    def _forward_impl(self, obs: torch.Tensor) -> KataGoOutput: ...
