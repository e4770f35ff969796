"""Transformer with learned row+column embeddings, scalar contract
(mirror of keisei/training/models/transformer.py:13-95; BASELINE config 5).  CPU tensors run the nn children; CUDA/HIP
tensors run the hand-written kernels of keisei_amd/hip/transformer.py (forward and backward)."""
from __future__ import annotations

import dataclasses

import torch
from torch import nn

from .base import BaseModel


@dataclasses.dataclass(frozen=True)
class TransformerParams:
    d_model: int
    nhead: int
    num_layers: int

    def __post_init__(self) -> None:
        if self.d_model <= 0:
            raise ValueError(f"d_model must be > 0, got {self.d_model}")
        if self.nhead <= 0:
            raise ValueError(f"nhead must be > 0, got {self.nhead}")
        if self.num_layers <= 0:
            raise ValueError(f"num_layers must be > 0, got {self.num_layers}")
        if self.d_model % self.nhead != 0:
            raise ValueError(f"d_model ({self.d_model}) must be divisible by nhead ({self.nhead})")


class TransformerModel(BaseModel):
    def __init__(self, params: TransformerParams) -> None:
        super().__init__()
        d, n = params.d_model, self.BOARD_SIZE
        self.input_proj = nn.Linear(self.OBS_CHANNELS, d)
        self.row_embed = nn.Embedding(n, d)
        self.col_embed = nn.Embedding(n, d)
        layer = nn.TransformerEncoderLayer(d_model=d, nhead=params.nhead, dim_feedforward=4 * d,
                                           batch_first=True, norm_first=True)
        self.encoder = nn.TransformerEncoder(layer, num_layers=params.num_layers, enable_nested_tensor=False)
        self.policy_fc = nn.Linear(d * n * n, self.ACTION_SPACE)
        self.value_fc1 = nn.Linear(d, d)
        self.value_fc2 = nn.Linear(d, 1)
        self.register_buffer("_row_idx", torch.arange(n), persistent=False)
        self.register_buffer("_col_idx", torch.arange(n), persistent=False)

    def forward(self, obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        self._check_obs(obs)
        if obs.is_cuda:
            # hand-written HIP path (keisei_amd/hip/transformer.py); raises if the library is missing -- no fallback
            from keisei_amd.hip.transformer import run_model
            return run_model(self, obs)
        b, sq = obs.shape[0], self.BOARD_SIZE * self.BOARD_SIZE
        tokens = self.input_proj(obs.permute(0, 2, 3, 1).reshape(b, sq, self.OBS_CHANNELS))
        pos = self.row_embed(self._row_idx)[:, None, :] + self.col_embed(self._col_idx)[None, :, :]
        tokens = self.encoder(tokens + pos.reshape(1, sq, -1))
        policy = self.policy_fc(tokens.reshape(b, -1))
        value = torch.tanh(self.value_fc2(torch.relu(self.value_fc1(tokens.mean(dim=1)))))
        return policy, value
