"""Value-head adapters (mirror of keisei/training/value_adapter.py:16-144).

Same class names, method signatures, validation messages and numerics: the scalar projection used
by GAE is P(W) - P(L) (optionally blended with clamp(score, -1, 1)), the multi-head loss is
lambda_value * CE(ignore_index=-1, all-ignored -> graph-connected zero) + lambda_score * MSE.
On the GPU training path KataGoPPOAlgorithm recognises MultiHeadValueAdapter and evaluates the
same formula inside the fused loss kernel (csrc/loss.hip) using the adapter's coefficients.
"""
from __future__ import annotations

import abc

import torch
import torch.nn.functional as F


class ValueHeadAdapter(abc.ABC):
    @abc.abstractmethod
    def scalar_value_from_output(self, value_output: torch.Tensor) -> torch.Tensor:
        """(batch,) scalar value for GAE."""

    @abc.abstractmethod
    def compute_value_loss(self, value_output, returns, value_cats, score_targets, score_pred=None) -> torch.Tensor:
        """Value loss for the model contract."""

    def scalar_value_blended(self, value_logits: torch.Tensor, score_lead: torch.Tensor) -> torch.Tensor:
        return self.scalar_value_from_output(value_logits)


class ScalarValueAdapter(ValueHeadAdapter):
    """tanh scalar value, MSE against returns."""

    def scalar_value_from_output(self, value_output: torch.Tensor) -> torch.Tensor:
        return value_output.squeeze(-1)

    def compute_value_loss(self, value_output, returns, value_cats=None, score_targets=None, score_pred=None):
        if returns is None:
            raise ValueError("ScalarValueAdapter requires returns")
        return F.mse_loss(value_output.squeeze(-1), returns)


class MultiHeadValueAdapter(ValueHeadAdapter):
    """W/D/L cross-entropy + score MSE."""

    def __init__(self, lambda_value: float = 1.5, lambda_score: float = 0.02, score_blend_alpha: float = 0.0) -> None:
        if lambda_value < 0:
            raise ValueError(f"lambda_value must be >= 0, got {lambda_value}")
        if lambda_score < 0:
            raise ValueError(f"lambda_score must be >= 0, got {lambda_score}")
        if not 0.0 <= score_blend_alpha <= 1.0:
            raise ValueError(f"score_blend_alpha must be in [0, 1], got {score_blend_alpha}")
        self.lambda_value = lambda_value
        self.lambda_score = lambda_score
        self.score_blend_alpha = score_blend_alpha

    def scalar_value_from_output(self, value_output: torch.Tensor) -> torch.Tensor:
        probs = torch.softmax(value_output, dim=-1)
        return probs[:, 0] - probs[:, 2]

    def scalar_value_blended(self, value_logits: torch.Tensor, score_lead: torch.Tensor) -> torch.Tensor:
        wdl = self.scalar_value_from_output(value_logits)
        a = self.score_blend_alpha
        if a == 0.0:
            return wdl
        return (1 - a) * wdl + a * score_lead.squeeze(-1).clamp(-1, 1)

    def compute_value_loss(self, value_output, returns=None, value_cats=None, score_targets=None, score_pred=None):
        if value_cats is None:
            raise ValueError("MultiHeadValueAdapter requires value_cats")
        if score_targets is None:
            raise ValueError("MultiHeadValueAdapter requires score_targets")
        if score_pred is None:
            raise ValueError("MultiHeadValueAdapter requires score_pred")
        if bool((value_cats >= 0).any()):
            ce = F.cross_entropy(value_output, value_cats, ignore_index=-1)
        else:
            ce = value_output.sum() * 0.0          # keeps the graph connected, gradient exactly zero
        mse = F.mse_loss(score_pred.squeeze(-1), score_targets)
        return self.lambda_value * ce + self.lambda_score * mse


def get_value_adapter(model_contract: str, lambda_value: float = 1.5, lambda_score: float = 0.02,
                      score_blend_alpha: float = 0.0) -> ValueHeadAdapter:
    if model_contract == "scalar":
        return ScalarValueAdapter()
    if model_contract == "multi_head":
        return MultiHeadValueAdapter(lambda_value=lambda_value, lambda_score=lambda_score,
                                     score_blend_alpha=score_blend_alpha)
    raise ValueError(f"Unknown model contract: {model_contract}")
