"""Algorithm registry (mirror of keisei/training/algorithm_registry.py:11-40): only ``katago_ppo`` is
trainable; the orphan ``PPOParams`` dataclass is kept because callers import it."""
from __future__ import annotations

import dataclasses
from typing import Any

from keisei_amd.training.katago_ppo import KataGoPPOParams


@dataclasses.dataclass(frozen=True)
class PPOParams:
    learning_rate: float = 3e-4
    gamma: float = 0.99
    clip_epsilon: float = 0.2
    epochs_per_batch: int = 4
    batch_size: int = 256
    entropy_coeff: float = 0.01
    value_loss_coeff: float = 0.5


_PARAM_SCHEMAS: dict[str, type] = {"katago_ppo": KataGoPPOParams}
VALID_ALGORITHMS = set(_PARAM_SCHEMAS)


def validate_algorithm_params(algorithm: str, params: dict[str, Any]) -> object:
    if algorithm not in _PARAM_SCHEMAS:
        raise ValueError(f"Unknown algorithm '{algorithm}'. Valid: {sorted(VALID_ALGORITHMS)}")
    try:
        return _PARAM_SCHEMAS[algorithm](**params)
    except TypeError as e:
        raise TypeError(f"Invalid params for '{algorithm}': {e}") from e
