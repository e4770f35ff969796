"""Architecture registry (mirror of keisei/training/model_registry.py:17-100): name ->
(model class, params dataclass, value contract, observation channels), with the reference's
validation behaviour (ValueError naming the offending field, TypeError for unknown/missing keys)."""
from __future__ import annotations

from typing import Any, NamedTuple

from torch import nn

from keisei_amd.training.models.mlp import MLPModel, MLPParams
from keisei_amd.training.models.resnet import ResNetModel, ResNetParams
from keisei_amd.training.models.se_resnet import SEResNetModel, SEResNetParams
from keisei_amd.training.models.transformer import TransformerModel, TransformerParams


class ArchitectureSpec(NamedTuple):
    model_cls: type[nn.Module]
    params_cls: type
    contract: str          # "scalar" | "multi_head"
    obs_channels: int


_REGISTRY: dict[str, ArchitectureSpec] = {
    "resnet": ArchitectureSpec(ResNetModel, ResNetParams, "scalar", 50),
    "mlp": ArchitectureSpec(MLPModel, MLPParams, "scalar", 50),
    "transformer": ArchitectureSpec(TransformerModel, TransformerParams, "scalar", 50),
    "se_resnet": ArchitectureSpec(SEResNetModel, SEResNetParams, "multi_head", 50),
}
VALID_ARCHITECTURES = set(_REGISTRY)


def _spec(architecture: str) -> ArchitectureSpec:
    try:
        return _REGISTRY[architecture]
    except KeyError:
        raise ValueError(f"Unknown architecture '{architecture}'. Valid: {sorted(VALID_ARCHITECTURES)}") from None


def _semantic_checks(architecture: str, v: Any) -> None:
    if architecture == "transformer":
        if v.nhead <= 0:
            raise ValueError(f"transformer: nhead must be > 0, got {v.nhead}")
        if v.d_model <= 0:
            raise ValueError(f"transformer: d_model must be > 0, got {v.d_model}")
        if v.d_model % v.nhead != 0:
            raise ValueError(f"transformer: d_model ({v.d_model}) must be divisible by nhead ({v.nhead})")
    elif architecture == "se_resnet":
        if v.channels <= 0:
            raise ValueError(f"se_resnet: channels must be > 0, got {v.channels}")
        if v.se_reduction <= 0:
            raise ValueError(f"se_resnet: se_reduction must be > 0, got {v.se_reduction}")
        if v.channels // v.se_reduction < 1:
            raise ValueError(f"se_resnet: channels ({v.channels}) // se_reduction ({v.se_reduction}) must be >= 1")
    elif architecture == "resnet":
        if v.hidden_size <= 0:
            raise ValueError(f"resnet: hidden_size must be > 0, got {v.hidden_size}")
        if v.num_layers < 0:
            raise ValueError(f"resnet: num_layers must be >= 0, got {v.num_layers}")
    elif architecture == "mlp":
        if any(s <= 0 for s in v.hidden_sizes):
            raise ValueError(f"mlp: all hidden_sizes must be > 0, got {v.hidden_sizes}")


def validate_model_params(architecture: str, params: dict[str, Any]) -> object:
    spec = _spec(architecture)
    try:
        validated = spec.params_cls(**params)
    except TypeError as e:
        raise TypeError(f"Invalid params for '{architecture}': {e}") from e
    _semantic_checks(architecture, validated)
    return validated


def build_model(architecture: str, params: dict[str, Any]) -> nn.Module:
    validated = validate_model_params(architecture, params)
    return _spec(architecture).model_cls(validated)


def get_model_contract(architecture: str) -> str:
    return _spec(architecture).contract


def get_obs_channels(architecture: str) -> int:
    return _spec(architecture).obs_channels
