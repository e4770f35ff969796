"""Which training algorithms exist and how their hyper-parameters are checked.

API mirror of keisei/training/algorithm_registry.py:11-40 -- ``VALID_ALGORITHMS``, ``PPOParams`` and
``validate_algorithm_params`` keep the reference's names, defaults and exception types.  Only ``katago_ppo`` is
trainable; ``PPOParams`` (the removed scalar-PPO trainer's knobs) survives because callers still import it.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Mapping

from keisei_amd.training.katago_ppo import KataGoPPOParams

# (field, type, default) of the legacy scalar-PPO parameter set
_LEGACY_PPO_FIELDS = (
    ("learning_rate", float, 3e-4),
    ("gamma", float, 0.99),
    ("clip_epsilon", float, 0.2),
    ("epochs_per_batch", int, 4),
    ("batch_size", int, 256),
    ("entropy_coeff", float, 0.01),
    ("value_loss_coeff", float, 0.5),
)
PPOParams = dataclasses.make_dataclass(
    "PPOParams", [(name, tp, dataclasses.field(default=default)) for name, tp, default in _LEGACY_PPO_FIELDS], frozen=True)
PPOParams.__module__ = __name__

_PARAM_SCHEMAS: dict[str, type] = {"katago_ppo": KataGoPPOParams}
VALID_ALGORITHMS = set(_PARAM_SCHEMAS)


def validate_algorithm_params(algorithm: str, params: Mapping[str, Any]) -> object:
    """Instantiate the parameter dataclass of ``algorithm``; unknown names are a ValueError, unknown or badly typed
    keys surface as the dataclass constructor's TypeError (range checks live in the dataclass itself)."""
    schema = _PARAM_SCHEMAS.get(algorithm)
    if schema is None:
        raise ValueError(f"Unknown algorithm '{algorithm}'. Valid: {sorted(VALID_ALGORITHMS)}")
    try:
        return schema(**params)
    except TypeError as err:
        raise TypeError(f"Invalid params for '{algorithm}': {err}") from err
